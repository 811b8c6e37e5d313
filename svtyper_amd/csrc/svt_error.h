// svt_error.h -- thread-local error text shared by every translation unit of libsvtyper_hip.so
#ifndef SVT_ERROR_H
#define SVT_ERROR_H

#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/svtyper_hip.h"

namespace svt {

inline thread_local std::string g_err;   // returned by svt_last_error()

inline int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

// The body of a C-ABI entry point: no C++ exception leaves the library (an allocation failure inside a
// container would otherwise terminate the caller's process).
template <typename F>
int guarded(F&& f) noexcept
{
    try {
        return f();
    } catch (const std::bad_alloc&) {
        try { return fail(SVT_ERR_NOMEM, "out of host memory"); } catch (...) { return SVT_ERR_NOMEM; }
    } catch (const std::exception& e) {
        try { return fail(SVT_ERR_INTERNAL, std::string("unexpected failure: ") + e.what()); } catch (...) { return SVT_ERR_INTERNAL; }
    } catch (...) {
        return SVT_ERR_INTERNAL;
    }
}

// fn(t) for t in [0, nt): t = 0 on the calling thread, the others on threads of their own.  An exception in any
// of them is rethrown here once all have finished (so it reaches guarded() instead of terminating the process);
// the share of a thread that could not be started runs on the calling thread.
template <typename Fn>
void run_threads(unsigned nt, Fn&& fn)
{
    std::exception_ptr first;
    std::mutex lock;
    auto body = [&](unsigned t) {
        try {
            fn(t);
        } catch (...) {
            std::lock_guard<std::mutex> g(lock);
            if (!first) first = std::current_exception();
        }
    };
    if (nt <= 1) {
        body(0);
    } else {
        // thread t starts threads 2t+1 and 2t+2 before it runs its own share: sixty-four threads started one after the other
        // by the caller cost ~1.5 ms, a sixth of a small call; as a tree they are all running after six generations
        std::function<void(unsigned)> node = [&](unsigned t) {
            std::thread kid[2];
            bool up[2] = {false, false};
            for (unsigned k = 0; k < 2 && 2 * t + 1 + k < nt; ++k) {
                try {
                    kid[k] = std::thread(node, 2 * t + 1 + k);
                    up[k] = true;
                } catch (const std::system_error&) {
                }
            }
            body(t);
            for (unsigned k = 0; k < 2 && 2 * t + 1 + k < nt; ++k) {
                if (up[k]) kid[k].join();
                else node(2 * t + 1 + k);      // could not be started: its share (and its children's) runs here
            }
        };
        node(0);
    }
    if (first) std::rethrow_exception(first);
}

}  // namespace svt

#endif  // SVT_ERROR_H
