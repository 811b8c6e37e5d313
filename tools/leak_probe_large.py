#!/usr/bin/env python
"""The same for batches large enough that their record buffers are chunked virtual-memory mappings (svt_host_transfer.h):
create / pass / destroy with svt_trim between, sizes going up and down; free device memory must come back every time."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from svtyper_amd import hip
rt = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    rt.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 1e6
big = bench.generate("c3_mixed_1m", 1_000_000, 0, bench.usable_cpus())
hip.load(); hip.device_count()
with hip.DeviceBatch(big.slice(0, 1000), 0, 0) as d:
    d.genotype(sync=True)
hip.trim()
base = free_mb()
print("free at start %.0f MB" % base)
want = None
for it, n in enumerate([1_000_000, 400_000, 1_000_000, 700_000, 350_000, 1_000_000, 900_000, 500_000] * 3):
    b = big.slice(0, n)
    with hip.DeviceBatch(b, 0, 0) as d:
        d.genotype(sync=True)
        r = d.results().rec[:1000].tobytes()
    if n == 1_000_000:
        want = want or r
        assert r == want
    held = base - free_mb()
    if it % 2 == 1:
        hip.trim()
    print("batch %2d (%7d units): %5.0f MB held before trim, %5.0f MB after%s" % (it, n, held, base - free_mb(), "" if it % 2 else " (not trimmed)"), flush=True)
hip.trim()
print("after the last svt_trim: %.0f MB held" % (base - free_mb()))
