#!/usr/bin/env python
"""Create / genotype / results / destroy many batches of varying size and watch the device's free memory:
the pools and caches must level off (svt_trim() gives everything back)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from svtyper_amd import hip, synth
rt = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    rt.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 1e6
lib = bench.fixture_library()
rng = np.random.default_rng(1)
hip.load(); hip.device_count()
base = free_mb()
print("free at start %.0f MB" % base)
for it in range(120):
    n = int(rng.integers(1000, 120000))
    b = synth.make_units(n, it, [lib], svtype_mix=(0.7, 0.15, 0.15, 0.0))
    with hip.DeviceBatch(b, 0, int(rng.integers(0, 4))) as d:
        d.genotype(sync=True); d.results()
    if it % 20 == 19:
        print("after %3d batches: free %.0f MB (%.0f MB held)" % (it + 1, free_mb(), base - free_mb()))
hip.trim()
print("after svt_trim: free %.0f MB (%.0f MB held)" % (free_mb(), base - free_mb()))
