"""svt_error.h: the exception guard of the C ABI and the thread runner -- compiled into a tiny host program with g++
(no GPU, no HIP) and run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r'''
#include <atomic>
#include <cstdio>
#include <stdexcept>
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
