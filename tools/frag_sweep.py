#!/usr/bin/env python
"""Pass time against the number of fragment records per unit (1 M units where memory allows):
   python tools/frag_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from svtyper_amd import hip, synth, evidence as ev

lib = bench.fixture_library()
for mean, n in ((2, 1_000_000), (10, 1_000_000), (30, 1_000_000), (100, 1_000_000), (300, 500_000), (1000, 150_000)):
    parts = [synth.make_units(n // 8, 100 + i, [lib], svtype_mix=(0.7, 0.15, 0.15, 0.0), mean_frags=mean,
                              sd_frags=max(1, mean // 4), min_frags=max(0, mean // 5), max_frags=mean * 2) for i in range(8)]
    b = ev.concat_batches(parts)
    for flags in (0, ev.FLAG_SSO_ASSOCIATION):
        with hip.DeviceBatch(b, 0, flags) as d:
            d.genotype(sync=True)
            ms = min(d.genotype_timed(10) / 10 for _ in range(3))
            alg, _ = d.bytes()
        print("F = %4d  units %8d  %s  pass %.4f ms  %.2f G units/s  %.2f TB/s algorithmic (frac %.3f)" % (
            mean, b.n_units, "sso    " if flags else "classic", ms, b.n_units / ms / 1e6, alg / ms / 1e9, alg / ms / 1e9 / 8))
