#!/usr/bin/env python
"""Kernel time per unit as a function of the batch size (tail / launch overhead vs steady state)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from svtyper_amd import hip
big = bench.generate("c3_mixed_1m", 4_000_000, 0, bench.usable_cpus())
for n in (250_000, 500_000, 1_000_000, 2_000_000, 4_000_000):
    b = big.slice(0, n)
    with hip.DeviceBatch(b, 0, 0) as d:
        d.genotype(sync=True)
        ms = min(d.genotype_timed(16) for _ in range(4)) / 16
        alg, res = d.bytes()
        print("%8d units: %.4f ms  %.2f ns/unit  resident %.0f MB  %.2f G units/s" % (n, ms, ms * 1e6 / n, res / 1e6, n / ms / 1e6))
