#!/usr/bin/env python
"""Kernel time of a configs[4]-shaped batch (sites x 32 samples, per-sample libraries) on one GPU."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svtyper_amd import synth, hip, evidence as ev
n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
t = time.time(); b = synth.make_multisample(n_sites, 32, seed=9); print("gen %.1fs units=%d records=%d libs=%d" % (time.time() - t, b.n_units, b.n_records, len(b.libs)))
for flags, name in ((0, "split"), (ev.FLAG_DENSE_LAYOUT, "dense")):
    d = hip.DeviceBatch(b, 0, flags)
    d.genotype(); ms = d.genotype_timed(20) / 20
    alg, res = d.bytes()
    print("%s: kernel %.4f ms  %.3e units/s  algorithmic %.0f GB/s  resident %.3f GB" % (name, ms, b.n_units / ms * 1e3, alg / ms / 1e6, res / 1e9))
    d.close()
