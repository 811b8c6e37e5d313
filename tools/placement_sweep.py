#!/usr/bin/env python
"""Where do the records and the result records have to lie in HBM for the pass to run at its fast level?  (DESIGN.md 3.1:
the same resident batch runs 0.315 or 0.336 ms depending on which physical blocks its two big buffers received.)

The real pass (svt_stream_kernel through the C ABI) over ONE 1 M-unit batch, with the two buffers placed by this tool
(svt_debug_device_alloc / svt_debug_bind_records / svt_batch_bind_device_results):
  A  result records at increasing byte offsets inside ONE physically contiguous chunk (same memory, another channel phase)
  B  records at increasing byte offsets inside ONE physically contiguous chunk
  C  a dozen separately allocated result buffers (hipMalloc), same records
  D  several separately allocated record buffers: hipMalloc, and virtual ranges over chunks of 2 MB ... 1 GB
  E  pairs: every record buffer of D against a few result buffers of C
Prints one line per placement: best and median pass time of a few groups of 20 back-to-back launches."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from svtyper_amd import hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
what = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else set("BCDE")
MB = 1 << 20

lib = hip.load()
lib.svt_debug_device_alloc.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
lib.svt_debug_device_free.argtypes = [C.c_int, C.c_void_p]
lib.svt_debug_record_bytes.restype = C.c_uint64
lib.svt_debug_record_bytes.argtypes = [C.c_void_p]
lib.svt_debug_bind_records.argtypes = [C.c_void_p, C.c_void_p]


def alloc(nbytes, chunk=0):
    p = C.c_void_p()
    hip._check(lib.svt_debug_device_alloc(0, int(nbytes), int(chunk), C.byref(p)))
    return int(p.value)


def free(p):
    hip._check(lib.svt_debug_device_free(0, C.c_void_p(p)))


batch = bench.generate("c3_mixed_1m", N, 0, bench.usable_cpus())
d = hip.DeviceBatch(batch, 0, 0)
d.genotype(sync=True)
want = d.results().rec.tobytes()
rec_bytes = int(lib.svt_debug_record_bytes(d._h))
res_bytes = N * 128
bench.spin_up(d, 60)


def timed(label, check=False):
    d.genotype(sync=True)
    ms = sorted(d.genotype_timed(20) / 20 for _ in range(5))
    ok = ""
    if check:
        ok = "  results %s" % ("equal" if d.results().rec.tobytes() == want else "DIFFER")
    print("%-64s %.4f ms (median %.4f)%s" % (label, ms[0], ms[2], ok), flush=True)
    return ms[0]


def bind_records(p):
    hip._check(lib.svt_debug_bind_records(d._h, C.c_void_p(p) if p else None))


timed("as created (pool buffers)", check=True)
timed("as created, again")

if "A" in what:
    print("# A: result records at byte offsets of one physically contiguous 1 GB chunk; records as created")
    big = alloc(1024 * MB, 1024 * MB)
    for off in [0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 128 << 10, 256 << 10, 512 << 10, MB, 2 * MB, 4 * MB, 8 * MB,
                16 * MB, 32 * MB, 64 * MB, 128 * MB, 256 * MB, 512 * MB, 0]:
        d.bind_device_results(big + off)
        timed("A results at chunk + %d" % off, check=off in (0, 4096))
    d.bind_device_results(0)
    free(big)

if "B" in what:
    print("# B: records at byte offsets of two 1 GB chunks; results as created")
    big = alloc(2048 * MB, 1024 * MB)
    for off in [0, 128, 1024, 4096, 16384, 65536, 256 << 10, MB, 2 * MB, 8 * MB, 32 * MB, 128 * MB, 256 * MB, 0]:
        bind_records(big + off)
        timed("B records at chunk + %d" % off, check=off in (0, 4096))
    bind_records(0)
    free(big)

res_bufs = []
if "C" in what or "E" in what:
    print("# C: separately allocated result buffers (hipMalloc of %d MB); records as created" % (res_bytes // MB))
    for i in range(12):
        p = alloc(res_bytes)
        res_bufs.append(p)
        d.bind_device_results(p)
        timed("C result buffer %2d at %#x" % (i, p))
    d.bind_device_results(0)

rec_bufs = []
if "D" in what or "E" in what:
    print("# D: separately allocated record buffers; results as created")
    for chunk in (0, 0, 0, 2 * MB, 32 * MB, 256 * MB, 256 * MB, 1024 * MB, 2048 * MB):
        try:
            p = alloc(rec_bytes, chunk)
        except hip.SvtyperHipError as e:
            print("chunk %d MB: %s" % (chunk // MB, e))
            continue
        rec_bufs.append((chunk, p))
        bind_records(p)
        timed("D records in %s at %#x" % ("one hipMalloc" if not chunk else "chunks of %d MB" % (chunk // MB), p), check=True)
    bind_records(0)

if "E" in what:
    print("# E: pairs (rows: record buffers of D, columns: result buffers 0-3 of C)")
    for chunk, p in rec_bufs:
        bind_records(p)
        row = []
        for q in res_bufs[:4]:
            d.bind_device_results(q)
            d.genotype(sync=True)
            row.append(min(d.genotype_timed(20) / 20 for _ in range(3)))
        print("E records %-16s %#x : %s" % ("hipMalloc" if not chunk else "chunks %d MB" % (chunk // MB), p, "  ".join("%.4f" % x for x in row)), flush=True)
    bind_records(0)
    d.bind_device_results(0)
if "G" in what:
    print("# G: full matrix, rows = record buffers, columns = 12 separately allocated result buffers (best of 3 x 20 launches)")
    res = [alloc(res_bytes) for _ in range(12)]
    recs = [("as created", 0)]
    for chunk in (0, 0, 2 * MB, 2 * MB, 256 * MB, 256 * MB, 1024 * MB):
        recs.append(("hipMalloc" if not chunk else "chunks %d MB" % (chunk // MB), alloc(rec_bytes, chunk)))
    for label, p in recs:
        bind_records(p)
        row = []
        for q in res:
            d.bind_device_results(q)
            d.genotype(sync=True)
            row.append(min(d.genotype_timed(20) / 20 for _ in range(3)))
        print("G %-16s %#14x : %s" % (label, p, " ".join("%.4f" % x for x in row)), flush=True)
    bind_records(0)
    d.bind_device_results(0)

if "F" in what:
    # is the fast level of a result buffer its residency in the 256 MB Infinity Cache (the same 128 MB of result lines are
    # rewritten by every pass of this benchmark)?  (1) rotate K result buffers from pass to pass; (2) evict everything
    # between passes with a 1 GB fill and time single passes
    lib.svt_debug_memset.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_uint64]
    print("# F: result buffers scanned again, then rotation / eviction")
    scan = []
    for i in range(16):
        p = alloc(res_bytes)
        d.bind_device_results(p)
        scan.append((timed("F result buffer %2d at %#x" % (i, p)), p))
    scan.sort()
    fast, slow = [p for _, p in scan[:4]], [p for _, p in scan[-4:]]

    def rotate(bufs, label):
        for q in bufs:
            d.bind_device_results(q)
            d.genotype(sync=True)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(40):
                d.bind_device_results(bufs[i % len(bufs)])
                d.genotype(sync=False)
            d.genotype(sync=True)
            dt = (time.perf_counter() - t0) / 41 * 1e3
            best = dt if best is None else min(best, dt)
        print("F rotate %-44s %.4f ms per pass (wall, 41 launches)" % (label, best), flush=True)
    for k in (1, 2, 3, 4):
        rotate(fast[:k], "%d fastest result buffer(s)" % k)
    for k in (1, 2, 4):
        rotate(slow[:k], "%d slowest result buffer(s)" % k)
    scratch = alloc(1024 * MB)
    for label, q in (("fastest", fast[0]), ("slowest", slow[0])):
        d.bind_device_results(q)
        d.genotype(sync=True)
        warm = d.genotype_timed(1)
        cold = []
        for _ in range(6):
            hip._check(lib.svt_debug_memset(0, C.c_void_p(scratch), 1, 1024 * MB))
            cold.append(d.genotype_timed(1))
        print("F single pass, %s result buffer: back to back %.4f ms; after a 1 GB fill: %s" % (label, warm, " ".join("%.4f" % x for x in sorted(cold))), flush=True)
    d.bind_device_results(0)
timed("as created, at the end", check=True)
