#!/usr/bin/env python
"""Launches of less than one round of resident workgroups: pass time against the number of units (20 k ... 250 k) and the
records per unit (1 ... 200), one library, both associations -- the fixed cost of a pass and the slope per 128-byte block
of a lane's unit (DESIGN.md 3.1 "small launches").
    python tools/small_launch.py [flags: r96]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from svtyper_amd import hip, synth, evidence as ev

r96 = ev.FLAG_RESULT96 if "r96" in sys.argv[1:] else 0
lib = bench.fixture_library()
for assoc, fl in (("classic", 0), ("sso", ev.FLAG_SSO_ASSOCIATION)):
    for n in (20_000, 50_000, 125_000, 250_000):
        row = []
        for mean in (1, 25, 50, 100, 200):
            b = synth.make_units(n, 100 + mean, [lib], svtype_mix=(0.7, 0.15, 0.15, 0.0), mean_frags=mean,
                                 sd_frags=max(1, mean // 4), min_frags=max(1, mean // 5), max_frags=mean * 2)
            with hip.DeviceBatch(b, 0, fl | r96) as d:
                d.genotype(sync=True)
                for _ in range(3):
                    d.genotype_timed(50)
                ms = min(d.genotype_timed(50) for _ in range(5)) / 50
            alg = 16 * b.n_records + 112 * b.n_units
            row.append("F=%3d %.4f ms (%.2f)" % (mean, ms, alg / ms / 1e6 / 8000))
        print("%-7s n %7d: %s" % (assoc, n, " | ".join(row)), flush=True)
