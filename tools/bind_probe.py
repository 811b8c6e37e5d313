#!/usr/bin/env python
"""Pass time of one resident batch writing its results (a) into the pool's own buffer, (b) into a torch tensor bound with
svt_batch_bind_device_results (what bench.py / distributed.py do for the gather), (c) into the pool's buffer wrapped as a torch
tensor.   python tools/bind_probe.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from svtyper_amd import hip, evidence as ev
b = bench.generate("c3_mixed_1m", 1_000_000, 0, bench.usable_cpus())
import torch
torch.cuda.set_device(0)
def t(d):
    d.genotype(sync=True); d.genotype_timed(100)
    return min(d.genotype_timed(10) / 10 for _ in range(8))
for rnd in range(3):
    with hip.DeviceBatch(b, 0, 0) as d:
        own = t(d)
        bufs = [torch.zeros(b.n_units * 128, dtype=torch.uint8, device="cuda") for _ in range(3)]
        times = []
        for x in bufs:
            d.bind_device_results(x.data_ptr()); times.append(t(d))
        d.bind_device_results(0)
        again = t(d)
    print("round %d: own buffer %.4f ms, three torch tensors %s, own again %.4f" % (rnd, own, " ".join("%.4f" % x for x in times), again), flush=True)
    del bufs
    hip.trim(); torch.cuda.empty_cache()
