#!/usr/bin/env python
"""Repeated svt_batch_create + genotype + results on the same host batch: end-to-end (PCIe-inclusive) rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from svtyper_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
batch = bench.generate("c3_mixed_1m", n, 0, len(os.sched_getaffinity(0)))
for it in range(4):
    t0 = time.perf_counter(); d = hip.DeviceBatch(batch, 0, 0); t1 = time.perf_counter()
    d.genotype(sync=True); t2 = time.perf_counter(); r = d.results(); t3 = time.perf_counter(); d.close()
    print("iter %d: create %.1f ms  genotype %.2f ms  results(D2H) %.1f ms  -> %.2f M breakpoints/s end to end" % (
        it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, n / (t3 - t0) / 1e6))
