// svt_error.h -- thread-local error text shared by every translation unit of libsvtyper_hip.so
#ifndef SVT_ERROR_H
#define SVT_ERROR_H

#include <pthread.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
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

// Parked worker threads of the host-side stages.  The calls that use run_threads() are short -- the reader of a chunk,
// its gather, the text of its sample columns: 1 to 40 ms -- and come one after the other, chunk after chunk.  Starting
// forty-seven threads took 3.5 ms on the 2 x EPYC 9575F box however they were started (one by one or as a tree: a new
// thread lands on an idle core that has to wake up first), more than the work of a small call; parked threads are all
// woken by one notify.  The pool serves one run_threads() at a time: a second caller (another host thread of the drivers'
// pipeline) finds it busy and starts threads of its own, as before.  Threads are created on demand (at most kMax), never
// destroyed; after a fork() the child starts with an empty pool.
class WorkerPool {
public:
    static WorkerPool& get()
    {
        static std::once_flag once;
        std::call_once(once, [] {
            instance().store(new WorkerPool(), std::memory_order_release);
            pthread_atfork(nullptr, nullptr, [] { instance().store(new WorkerPool(), std::memory_order_release); });   // (the parent's threads are not in the child)
        });
        return *instance().load(std::memory_order_acquire);
    }
    // body(t) for t in [1, nt) on parked threads and body(0) here; false (nothing has run) when the pool is taken or too small
    bool try_run(unsigned nt, const std::function<void(unsigned)>& body)
    {
        if (nt < 2 || nt - 1 > kMax) return false;
        if (taken_.exchange(true, std::memory_order_acquire)) return false;      // (also a run_threads() inside a share of another)
        struct Release {
            std::atomic<bool>& flag;
            ~Release() { flag.store(false, std::memory_order_release); }
        } release{taken_};
        const unsigned want = nt - 1;
        {
            std::lock_guard<std::mutex> g(lock_);
            while (threads_ < want) {
                try {
                    std::thread(&WorkerPool::park, this, threads_, generation_).detach();
                } catch (const std::system_error&) {
                    return false;
                }
                ++threads_;
            }
            job_ = &body;
            job_threads_ = want;
            remaining_ = want;
            ++generation_;
        }
        wake_.notify_all();
        body(0);
        std::unique_lock<std::mutex> g(lock_);
        done_.wait(g, [&] { return remaining_ == 0; });
        job_ = nullptr;
        job_threads_ = 0;
        return true;
    }

private:
    static constexpr unsigned kMax = 96;
    static std::atomic<WorkerPool*>& instance()
    {
        static std::atomic<WorkerPool*> p{nullptr};
        return p;
    }
    void park(unsigned index, uint64_t seen)
    {
        std::unique_lock<std::mutex> g(lock_);
        for (;;) {
            wake_.wait(g, [&] { return generation_ != seen; });
            seen = generation_;
            if (index < job_threads_) {
                const std::function<void(unsigned)>* job = job_;
                g.unlock();
                (*job)(index + 1);                         // (run_threads' body: it does not throw)
                g.lock();
                if (--remaining_ == 0) done_.notify_one();
            }
        }
    }
    std::atomic<bool> taken_{false};
    std::mutex lock_;
    std::condition_variable wake_, done_;
    const std::function<void(unsigned)>* job_ = nullptr;
    unsigned threads_ = 0, job_threads_ = 0, remaining_ = 0;
    uint64_t generation_ = 0;
};

// fn(t) for t in [0, nt): t = 0 on the calling thread, the others on parked threads of the pool (on threads of their own
// when the pool is taken).  An exception in any
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
    } else if (!WorkerPool::get().try_run(nt, std::function<void(unsigned)>(std::cref(body)))) {
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
