// svt_error.h -- thread-local error text shared by every translation unit of libsvtyper_hip.so
#ifndef SVT_ERROR_H
#define SVT_ERROR_H

#include <exception>
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
        std::vector<std::thread> pool;
        pool.reserve(nt - 1);
        unsigned started = 1;
        for (; started < nt; ++started) {
            try {
                pool.emplace_back(body, started);
            } catch (const std::system_error&) {
                break;
            }
        }
        body(0);
        for (unsigned t = started; t < nt; ++t) body(t);
        for (auto& th : pool) th.join();
    }
    if (first) std::rethrow_exception(first);
}

}  // namespace svt

#endif  // SVT_ERROR_H
