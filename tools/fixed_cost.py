#!/usr/bin/env python
"""What a pass costs when there is almost nothing to stream: 1 M units with ~1 record each (launch, table
staging, epilogue and the 128 MB of result records), then the same with the usual ~100 records."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from svtyper_amd import hip, synth, evidence as ev
lib = bench.fixture_library()
for mean in (1, 10, 100):
    parts = [synth.make_units(125_000, 100 + i, [lib], svtype_mix=(0.7, 0.15, 0.15, 0.0), mean_frags=mean,
                              sd_frags=max(1, mean // 4), min_frags=max(1, mean // 5), max_frags=mean * 2) for i in range(8)]
    b = ev.concat_batches(parts)
    with hip.DeviceBatch(b, 0, 0) as d:
        d.genotype(sync=True)
        ms = min(d.genotype_timed(16) for _ in range(4)) / 16
        alg, res = d.bytes()
        print("mean %3d records/unit: %d units %d records: %.4f ms, resident %.0f MB + results %.0f MB" % (
            mean, b.n_units, b.n_records, ms, res / 1e6, b.n_units * 128 / 1e6))

# the same near-empty pass at several sizes: t = launch constant + per-unit cost
print("near-empty pass (1 record per unit) by size:")
for n_parts in (1, 2, 4, 8, 16):
    parts = [synth.make_units(125_000, 300 + i, [lib], svtype_mix=(0.7, 0.15, 0.15, 0.0), mean_frags=1, sd_frags=1,
                              min_frags=1, max_frags=2) for i in range(n_parts)]
    b = ev.concat_batches(parts)
    with hip.DeviceBatch(b, 0, 0) as d:
        d.genotype(sync=True)
        ms = min(d.genotype_timed(16) for _ in range(4)) / 16
        print("  %8d units: %.4f ms" % (b.n_units, ms))
