#!/usr/bin/env python
"""Timing-only ablations of the per-unit part of the pass (libraries built with -DSVT_ABL_x under csrc/variants;
their results are wrong on purpose): near-empty batch (1 record per unit) and the headline shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from svtyper_amd import hip, synth, evidence as ev
lib = bench.fixture_library()
for mean in (1, 100):
    parts = [synth.make_units(125_000, 100 + i, [lib], svtype_mix=(0.7, 0.15, 0.15, 0.0), mean_frags=mean,
                              sd_frags=max(1, mean // 4), min_frags=max(1, mean // 5), max_frags=mean * 2) for i in range(8)]
    b = ev.concat_batches(parts)
    with hip.DeviceBatch(b, 0, 0) as d:
        d.genotype(sync=True)
        ms = min(d.genotype_timed(16) for _ in range(4)) / 16
        print("%s mean %3d: %.4f ms" % (os.path.basename(os.environ.get("SVTYPER_HIP_LIB", "default")), mean, ms))
