"""The host side of libsvtyper_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5 asked
the new build for sanitizer runs of its native code; the reference has none).  `make -C svtyper_amd/csrc asan` builds
the library once more with -fsanitize=address,undefined on the host code -- svt_reads.cpp parses untrusted BAM / BGZF /
BAI bytes, svt_pack.cpp and svt_format.cpp index caller arrays -- and the host-side test files plus a bounded corpus of
corrupted BAMs (tools/fuzz_bam.py) run against it in a subprocess with the sanitizer runtime preloaded.  A finding
aborts that process (halt_on_error), which fails the test.  CPU only: no device is involved."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svtyper_amd", "csrc")
ASAN_LIB = os.path.join(CSRC, "variants", "libsvtyper_hip_asan.so")


@pytest.fixture(scope="module")
def asan_env():
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc / clang sanitizer runtime in this image")
    r = subprocess.run(["make", "-s", "-C", CSRC, "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and os.path.exists(ASAN_LIB), r.stderr[-3000:]
    syms = subprocess.run(["nm", "-D", ASAN_LIB], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__ubsan_handle" in syms, "the library is not instrumented"
    return dict(os.environ, LD_PRELOAD=rt[-1], SVTYPER_HIP_LIB=ASAN_LIB, SVT_ALLOW_NO_GPU="1",
                ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def test_the_sanitizer_reports_through_this_setup(asan_env):
    """a deliberate heap overflow inside an instrumented call must kill the child: proves the runtime is live"""
    code = ("import ctypes as C, os\n"
            "L = C.CDLL(os.environ['SVTYPER_HIP_LIB'])\n"
            "buf = (C.c_uint8 * 64)()\n"      # 64 bytes where svt_results_host_sq expects 2 * 128
            "L.svt_results_host_sq.argtypes = [C.c_void_p, C.c_uint64]\n"
            "import ctypes.util\n"
            "libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; p = libc.malloc(64)\n"
            "L.svt_results_host_sq(C.c_void_p(p), 2)\n"
            "print('survived')\n")
    r = subprocess.run([sys.executable, "-c", code], env=asan_env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "AddressSanitizer" in r.stderr and "survived" not in r.stdout, (r.returncode, r.stderr[-500:])


def test_host_side_tests_under_asan_and_ubsan(asan_env):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        "tests/test_native_reads.py", "tests/test_packed_evidence.py", "tests/test_cpp_helpers.py",
                        "tests/test_host_entries.py", "tests/test_bulk_vcf.py"], cwd=ROOT, env=asan_env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert " passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_corrupted_bams_under_asan_and_ubsan(asan_env):
    env = dict(asan_env, SVT_FUZZ_ITERS="18")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_bam.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "crash" not in r.stdout and ("ok" in r.stdout or "error" in r.stdout), r.stdout[-1000:]


def test_packed_encoder_under_thread_sanitizer(tmp_path):
    """svt_pack.cpp hands chunks out through atomic counters and runs both of its phases on one set of threads with a
    hand-written barrier between them: a data race there would be silent.  tests/native/tsan_pack_main.cpp drives it
    (success and error path) with eight workers under -fsanitize=thread."""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "tsan_pack")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "native", "tsan_pack_main.cpp"), os.path.join(CSRC, "svt_pack.cpp"), "-o", exe,
                        "-lpthread"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "tsan" in r.stderr.lower():
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], env=dict(os.environ, SVT_PACK_THREADS="8", TSAN_OPTIONS="halt_on_error=1"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.stdout[-1000:], r.stderr[-3000:])
    lines = r.stdout.strip().splitlines()
    ranged, lines = lines[:2], lines[2:]
    # the ranged form: every unit and slot handed over exactly once, in ten ranges; a refused range stops the call with its code
    assert ranged[0].startswith("ranged 0 rc 0 units 20000 ") and ranged[0].endswith("calls 10"), ranged
    slots, of = (int(x) for x in ranged[0].split(" slots ")[1].split(" calls ")[0].split(" of "))
    assert slots == of > 0
    assert ranged[1].startswith("ranged 1 rc -7 units 4096 "), ranged
    assert len(lines) == 5 and all(" rc 0 " in l for l in lines[:3]) and all("rec_offset not monotone" in l for l in lines[3:]), lines


def test_native_reader_under_thread_sanitizer(tmp_path):
    """svt_reads.cpp: the workers of svt_bam_summarise claim runs of units from one counter and share inflated BGZF blocks
    (SharedBlocks: a mutex per shard, blocks immutable once published).  tests/native/tsan_reads_main.cpp runs the fixture's
    sites (x 3, so that every block is wanted by several workers) with 1 and 8 workers under -fsanitize=thread; the summaries
    must be the same bytes whatever the number of workers."""
    import json
    import shutil
    import numpy as np
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "tsan_reads")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "native", "tsan_reads_main.cpp"), os.path.join(CSRC, "svt_reads.cpp"), "-o", exe,
                        "-lz", "-ldl", "-lpthread"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "libtsan" in r.stderr.lower():
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    assert r.returncode == 0, r.stderr[-3000:]
    # the units exactly as NativeUnitCollector hands them to the ctypes layer
    sys.path.insert(0, ROOT)
    from svtyper_amd import bam as pybam, library, native_reads as nr, pipeline
    from svtyper_amd.vcf import Variant, Vcf
    data = os.path.join(ROOT, "tests", "data")
    bam_path = os.path.join(data, "NA12878.target_loci.sorted.bam")
    vcf, sites = Vcf(), []
    with open(os.path.join(data, "example.vcf")) as f:
        lines = f.readlines()
    vcf.add_header([l for l in lines if l.startswith("##")])
    for line in lines:
        if not line.startswith("#"):
            v = Variant(line.rstrip().split("\t"), vcf)
            bp = vcf.get_variant_breakpoints(v, 1e10) if v.has_svtype() and v.is_valid_svtype() else None
            if bp is not None:
                sites.append(bp)
    with open(os.path.join(data, "NA12878.bam.json")) as f:
        sample = library.Sample.from_lib_info(pybam.AlignmentFile(bam_path), json.load(f), 1e-3)

    class Capture:
        """stands where nr.NativeBam stands: keeps what summarise() is handed"""
        filename, lengths = bam_path, pybam.AlignmentFile(bam_path).lengths
        gettid = staticmethod(pybam.AlignmentFile(bam_path).gettid)

        def summarise(self, win, bps, rgs, idx, max_reads, mode, n_threads):
            win.tofile(str(tmp_path / "win.bin"))
            bps.tofile(str(tmp_path / "bps.bin"))
            self.rgs = ["%s=%d" % (rg, lib) for rg, lib in zip(rgs, idx)]
            raise StopIteration

    cap = Capture()
    coll = pipeline.NativeUnitCollector([sample], [cap], 1.0, 1.0, 20, nr.COUNT_SSO, 1000, geometry="device")
    for _ in range(3):
        for bp in sites:
            coll.add_site(bp)
    engine = type("E", (), {"genotype_fragments": lambda self, *a, **k: None})()
    with pytest.raises(StopIteration):
        coll.run(engine, 0)
    r = subprocess.run([exe, bam_path, str(tmp_path / "win.bin"), str(tmp_path / "bps.bin")] + cap.rgs,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.stdout[-1000:], r.stderr[-3000:])
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 6 and lines[0].startswith("threads 1 units %d " % (3 * len(sites))), lines
    assert len({l.split(" units ")[1] for l in lines[:3]}) == 1, lines      # summaries: same fragments, same bytes
    assert len({l.split(" units ")[1] for l in lines[3:]}) == 1, lines      # evidence records (svt_bam_evidence): the same
    assert all(l.startswith("evidence threads ") for l in lines[3:])
    assert int(lines[0].split(" fragments ")[1].split()[0]) > 10_000
