#!/usr/bin/env python
"""Per-launch time of the genotype kernel as a function of the number of back-to-back launches between
two HIP events (separates kernel duration from the inter-launch gap)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from svtyper_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
batch = bench.generate("c3_mixed_1m", n, 0, bench.usable_cpus())
for flags, name in ((0, "compact"), (2, "dense")):
    with hip.DeviceBatch(batch, 0, flags) as d:
        d.genotype(sync=True)
        for k in (1, 1, 2, 4, 8, 16, 32, 64):
            ms = min(d.genotype_timed(k) for _ in range(5))
            print("%s: %2d launches between events: %.4f ms total, %.4f ms per launch" % (name, k, ms, ms / k))
