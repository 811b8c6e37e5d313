"""svt_error.h: the exception guard of the C ABI and the thread runner (with its pool of parked threads) -- compiled into a tiny host program with g++
(no GPU, no HIP) and run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r'''
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include "svtyper_amd/csrc/svt_error.h"
using namespace svt;
int main()
{
    // 1. exceptions become error codes + a message, nothing escapes
    if (guarded([]() -> int { throw std::bad_alloc(); }) != SVT_ERR_NOMEM) return 1;
    if (guarded([]() -> int { throw std::runtime_error("boom"); }) != SVT_ERR_INTERNAL) return 2;
    if (g_err.find("boom") == std::string::npos) return 3;
    if (guarded([]() -> int { throw 42; }) != SVT_ERR_INTERNAL) return 4;
    if (guarded([]() -> int { return 7; }) != 7) return 5;
    // 2. every share runs exactly once, on 1 or many threads
    for (unsigned nt : {1u, 2u, 7u, 64u}) {
        std::atomic<unsigned> sum(0), calls(0);
        run_threads(nt, [&](unsigned t) { sum += t + 1; ++calls; });
        if (calls != nt || sum != nt * (nt + 1) / 2) return 6;
    }
    // 3. an exception in a worker reaches the caller after the others have finished
    std::atomic<unsigned> finished(0);
    const int rc = guarded([&]() -> int {
        run_threads(8, [&](unsigned t) {
            if (t == 3) throw std::runtime_error("worker 3");
            ++finished;
        });
        return 0;
    });
    if (rc != SVT_ERR_INTERNAL || finished != 7 || g_err.find("worker 3") == std::string::npos) return 7;
    // 4. the parked pool: call after call (growing and shrinking), from two callers at once (one finds the pool taken and
    //    starts threads of its own), a call inside a share, and in a forked child (which has none of the parent's threads)
    for (unsigned rep = 0; rep < 300; ++rep) {
        const unsigned nt = 2 + rep % 40;
        std::atomic<unsigned> sum(0);
        run_threads(nt, [&](unsigned t) { sum += t + 1; });
        if (sum != nt * (nt + 1) / 2) return 8;
    }
    std::atomic<unsigned> bad(0);
    auto caller = [&] {
        for (unsigned rep = 0; rep < 200; ++rep) {
            std::atomic<unsigned> sum(0);
            run_threads(9, [&](unsigned t) { sum += t + 1; });
            if (sum != 45) ++bad;
        }
    };
    std::thread other(caller);
    caller();
    other.join();
    if (bad) return 9;
    std::atomic<unsigned> inner(0);
    run_threads(4, [&](unsigned) { run_threads(3, [&](unsigned t) { inner += t + 1; }); });
    if (inner != 4 * 6) return 10;
#ifndef SVT_NO_FORK_CHECK
    const pid_t child = fork();
    if (child == 0) {
        std::atomic<unsigned> sum(0);
        run_threads(12, [&](unsigned t) { sum += t + 1; });
        _exit(sum == 78 ? 0 : 1);
    }
    int status = 0;
    if (child < 0 || waitpid(child, &status, 0) != child || !WIFEXITED(status) || WEXITSTATUS(status) != 0) return 11;
#endif
    std::puts("ok");
    return 0;
}
'''


def test_exception_guard_and_thread_runner(tmp_path):
    src = tmp_path / "helpers.cpp"
    src.write_text(PROGRAM)
    exe = str(tmp_path / "helpers")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I", ROOT, str(src), "-o", exe], check=True, timeout=120)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout, r.stderr)


def test_thread_runner_under_thread_sanitizer(tmp_path):
    """the same program under -fsanitize=thread (without the fork() check, which that runtime does not support): the hand-over
    of a job to the parked threads and back is two condition variables and a generation counter"""
    import pytest
    src = tmp_path / "helpers.cpp"
    src.write_text(PROGRAM)
    exe = str(tmp_path / "helpers_tsan")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-DSVT_NO_FORK_CHECK", "-pthread", "-I", ROOT, str(src), "-o", exe],
                       capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "libtsan" in r.stderr.lower():
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}      # (this file also runs under a preloaded ASan runtime)
    r = subprocess.run([exe], env=dict(env, TSAN_OPTIONS="halt_on_error=1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok" and "ThreadSanitizer" not in r.stderr, (r.returncode, r.stdout, r.stderr[-3000:])


FORMAT_PROGRAM = r'''
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include "svtyper_amd/csrc/svt_fast_format.h"
static long checked = 0;
static int check(double v)
{
    char a[64], b[64];
    for (int d : {0, 2}) {
        const int n = svt::format_fixed(a, v, d);
        if (n == 0) { if (std::fabs(v) < 8e12) return 1; continue; }      // the fast range must be taken
        a[n] = 0;
        std::snprintf(b, sizeof b, d == 0 ? "%.0f" : "%0.2f", v);
        if (std::strcmp(a, b) != 0) { std::printf("fixed %d: %.17g -> '%s' vs '%s'\n", d, v, a, b); return 2; }
        ++checked;
    }
    const int n = svt::format_g2(a, v);
    if (n == 0) return (v == 0.0 && !std::signbit(v)) || (v >= 1e-4 && v <= 1.0) ? 3 : 0;
    a[n] = 0;
    std::snprintf(b, sizeof b, "%.2g", v);
    if (std::strcmp(a, b) != 0) { std::printf("g2: %.17g -> '%s' vs '%s'\n", v, a, b); return 4; }
    ++checked;
    return 0;
}
int main()
{
    // every allele balance qa / (qr + qa) of small counts, ties of both conversions, decade edges, signs, zeros
    for (int a = 0; a <= 400; ++a)
        for (int b = 0; b <= 400; ++b)
            if (a + b) if (int rc = check((double)a / (double)(a + b))) return rc;
    for (int k = -2000; k <= 2000; ++k)
        for (double f : {0.0, 0.5, 0.125, 0.375, 0.625, 0.875, 0.005, 0.015, 0.025, 0.045, 0.995, 0.9949999999999999, 0.49999999999999994})
            if (int rc = check(k + (k < 0 ? -f : f))) return rc;
    for (double v : {0.0, -0.0, 1e-4, 9.95e-5, 0.00010000000000000002, 0.000995, 0.00995, 0.0995, 0.995, 0.996, 0.9949, 1.0, 0.1, 0.01, 0.001,
                     0.09999999999999999, 0.009999999999999998, 5e-324, 2.2250738585072014e-308, 1e-300, -1e-300, 8.79e12, -8.79e12, 1e13, 1e300,
                     -1e300, 4503599627370496.5, 0.285, 1.005, 2.675, 1e-5, 3e-5, 1.5, 2.0, (double)INFINITY, -(double)INFINITY, (double)NAN})
        if (int rc = check(v)) return rc;
    std::mt19937_64 rng(7);
    for (int i = 0; i < 3000000; ++i) {
        uint64_t bits = rng();
        double v;
        std::memcpy(&v, &bits, 8);                                   // any double at all (most outside the fast range)
        if (int rc = check(v)) return rc;
        const double u = (double)(rng() >> 11) * 0x1p-53;            // [0, 1)
        for (double w : {u, -u * 700.0, u * 1e4, -u * 3e9, u * 1e-3, std::ldexp(u, -(int)(rng() % 80)), (double)(rng() % 100000) / 100.0 + 0.005})
            if (int rc = check(w)) return rc;
    }
    std::printf("ok %ld\n", checked);
    return 0;
}
'''


def test_fast_format_prints_what_printf_prints(tmp_path):
    """svt_fast_format.h ('%.0f', '%0.2f', '%.2g' by exact integer arithmetic) against snprintf: the text of the VCF sample
    columns (svtyper/parsers.py:391-399, classic.py:466-469) must not change with the way it is produced."""
    src = tmp_path / "fmt.cpp"
    src.write_text(FORMAT_PROGRAM)
    exe = str(tmp_path / "fmt")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", ROOT, str(src), "-o", exe], check=True, timeout=120)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), (r.returncode, r.stdout[-400:], r.stderr[-400:])
    assert int(r.stdout.split()[1]) > 20_000_000
