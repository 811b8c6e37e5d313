#!/usr/bin/env python
"""Does the pass time depend on where the batch's buffers land in HBM?  The same 1 M-unit batch is made resident several
times in one process: first through the device-buffer pool (the same allocations again), then with svt_trim between
(fresh hipMalloc each time), and the pass is timed each time.   python tools/alloc_variance.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from svtyper_amd import hip
b = bench.generate("c3_mixed_1m", 1_000_000, 0, bench.usable_cpus())
def once(tag):
    t0 = time.perf_counter()
    with hip.DeviceBatch(b, 0, 0) as d:
        create_ms = (time.perf_counter() - t0) * 1e3
        d.genotype(sync=True)
        d.genotype_timed(100)
        ms = sorted(d.genotype_timed(10) / 10 for _ in range(12))
    print("%-22s create %6.1f ms  pass %.4f ms (median %.4f)" % (tag, create_ms, ms[0], ms[len(ms) // 2]), flush=True)
for i in range(4):
    once("pooled buffers #%d" % i)
for i in range(6):
    hip.trim()
    once("fresh allocations #%d" % i)
