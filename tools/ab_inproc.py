#!/usr/bin/env python
"""A/B of library builds IN ONE PROCESS OVER THE SAME DEVICE MEMORY.  Where a batch's records and result records lie in HBM
moves the pass by up to 8 %, more than most kernel changes: builds compared in separate processes, each with
its own allocations, measure the allocator.  Here every build (svtyper_amd/csrc/libsvtyper_hip.so + variants/lib_*.so, each
dlopen'ed beside the others) makes the same batch resident and is then pointed at the FIRST build's record and result
buffers (svt_debug_bind_records / svt_batch_bind_device_results); the passes are timed round-robin.
    python tools/ab_inproc.py [workload: c3|sso|c5|c5site] [units] [variant name filters...]"""
import ctypes as C
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from svtyper_amd import evidence as ev, hip, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
r96 = wl.endswith("96")          # c396 / sso96 / c596: SVT_FLAG_RESULT96 (builds that do not know the flag are left out)
both = wl.endswith("both")       # c3both / ...: every build with 128-byte AND with 96-byte device records, over the same buffers
wl = wl[:-2] if r96 else wl[:-4] if both else wl
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
only = sys.argv[3:]
flags = (ev.FLAG_SSO_ASSOCIATION if wl == "sso" else 0) | (ev.FLAG_RESULT96 if r96 else 0)
if wl in ("c3", "sso"):
    batch = bench.generate("c3_mixed_1m", n, 0, bench.usable_cpus())
    order = 0
elif wl == "c5":
    batch = bench.generate("c5_multisample", n, 0, bench.usable_cpus(), layout="sample")
    order = 32
else:
    batch = bench.generate("c5_multisample", n, 0, bench.usable_cpus())
    order = 0
libs = [hip.LIB_PATH] + sorted(glob.glob(os.path.join(ROOT, "svtyper_amd", "csrc", "variants", "lib_*.so")))
if only:
    libs = libs[:1] + [l for l in libs[1:] if any(o in os.path.basename(l) for o in only)]


class Build:
    def __init__(self, path, flags=None):
        flags = globals()["flags"] if flags is None else flags
        self.name = os.path.basename(path) + (" r96" if flags & ev.FLAG_RESULT96 else "")
        L = self.L = C.CDLL(path)
        L.svt_last_error.restype = C.c_char_p
        L.svt_batch_create.argtypes = [C.POINTER(hip.CEvidenceBatch), C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
        L.svt_batch_genotype.argtypes = [C.c_void_p, C.c_int]
        L.svt_batch_genotype_timed.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.svt_batch_results.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.svt_batch_result_order.argtypes = [C.c_void_p, C.c_uint32]
        L.svt_batch_device_results.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.svt_batch_bind_device_results.argtypes = [C.c_void_p, C.c_void_p]
        L.svt_debug_records_ptr.restype = C.c_void_p
        L.svt_debug_records_ptr.argtypes = [C.c_void_p]
        L.svt_debug_bind_records.argtypes = [C.c_void_p, C.c_void_p]
        self.h = C.c_void_p()
        cb = batch.as_c()
        self.check(L.svt_batch_create(C.byref(cb), 0, flags, C.byref(self.h)))
        if order:
            self.check(L.svt_batch_result_order(self.h, order))

    def check(self, rc):
        if rc:
            raise RuntimeError("%s: %s" % (self.name, self.L.svt_last_error().decode()))

    def buffers(self):
        out = C.c_void_p()
        self.check(self.L.svt_batch_device_results(self.h, C.byref(out)))
        return int(self.L.svt_debug_records_ptr(self.h)), int(out.value)

    def bind(self, rec, out):
        self.check(self.L.svt_debug_bind_records(self.h, C.c_void_p(rec)))
        self.check(self.L.svt_batch_bind_device_results(self.h, C.c_void_p(out)))

    def timed(self, iters=20):
        ms = C.c_float()
        self.check(self.L.svt_batch_genotype_timed(self.h, iters, C.byref(ms)))
        return ms.value / iters

    def digest(self):
        self.check(self.L.svt_batch_genotype(self.h, 1))
        # (every build shares ONE result buffer: run this build's pass again right before reading it)
        r = ev.Results.empty(batch.n_units)
        self.check(self.L.svt_batch_results(self.h, C.c_void_p(r.ptr()), batch.n_units))
        return hashlib.sha1(r.rec.tobytes()).hexdigest()[:12]


builds = []
for p in libs:
    for fl in ([flags, flags | ev.FLAG_RESULT96] if both else [flags]):
        try:
            builds.append(Build(p, fl))
        except RuntimeError as e:
            print("left out: %s" % e)
rec, out = builds[0].buffers()
for b in builds[1:]:
    b.bind(rec, out)
alg = 16 * batch.n_records + 112 * batch.n_units
for b in builds:          # spin up
    for _ in range(5):
        b.timed(20)
times = {b.name: [] for b in builds}
for _ in range(int(os.environ.get("AB_ROUNDS", "8"))):
    for b in builds:
        times[b.name].append(b.timed(20))
print("%s%s: %d units, %d records, every build over the record / result buffers of %s" % (wl, " (96-byte result records)" if r96 else "", batch.n_units, batch.n_records, builds[0].name))
for b in builds:
    t = sorted(times[b.name])
    print("%-34s best %.4f ms  median %.4f  frac(best) %.3f  digest %s" % (b.name, t[0], t[len(t) // 2], alg / t[0] / 1e6 / 8000, b.digest()), flush=True)
