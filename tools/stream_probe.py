#!/usr/bin/env python
"""First look at the streaming layout: parity against the oracle on small batches, then kernel time at full size
next to the tiled layouts.   python tools/stream_probe.py [units]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from svtyper_amd import evidence as ev, hip, synth
from oracle import c_oracle

lib = bench.fixture_library()


def parity(name, batch, flags):
    got = hip.genotype_batch(batch, flags=flags)
    want = c_oracle.genotype_batch(batch, flags=flags & ev.FLAG_SSO_ASSOCIATION)
    ok = (np.array_equal(got.gt, want.gt) and np.array_equal(got.counts, want.counts)
          and np.array_equal(got.tallies.view(np.uint64), want.tallies.view(np.uint64))
          and np.array_equal(got.gl.view(np.uint64), want.gl.view(np.uint64))
          and float(np.max(np.abs(got.sq - want.sq), initial=0.0)) <= 1e-6)
    print("parity %-28s flags %2d: %s" % (name, flags, "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        bad = np.nonzero((got.gt != want.gt) | (got.counts != want.counts).any(axis=1)
                         | (got.tallies.view(np.uint64) != want.tallies.view(np.uint64)).any(axis=1))[0]
        print("   first bad units", bad[:8], "of", batch.n_units)
        for u in bad[:3]:
            print("   got ", got.rec[u]); print("   want", want.rec[u])
    return ok


if "--no-parity" not in sys.argv:
    S = ev.FLAG_STREAM_LAYOUT
    allok = True
    for fl in (S, S | ev.FLAG_SSO_ASSOCIATION):
        allok &= parity("edge cases", synth.make_edge_cases([lib], seed=11), fl)
        allok &= parity("c2 20k", synth.make_config("c2_del_100k", [lib], n_units=20_000), fl)
        allok &= parity("c3 30k", synth.make_config("c3_mixed_1m", [lib], n_units=30_000), fl)
        for n in (1, 63, 64, 65, 511, 513, 4097):
            allok &= parity("tiny %d" % n, synth.make_units(n, n, [lib], svtype_mix=(0.5, 0.2, 0.2, 0.1), min_frags=0,
                                                            mean_frags=30, sd_frags=30), fl)
        libs = [lib, synth.normal_library(420.0, 95.0, seed=3), synth.normal_library(280.0, 40.0, seed=4)]
        allok &= parity("3 libraries (general)", synth.make_units(5000, 99, libs, svtype_mix=(0.5, 0.2, 0.2, 0.1)), fl)
    print("PARITY", "GREEN" if allok else "RED", flush=True)

n = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 1_000_000
batch = bench.generate("c3_mixed_1m", n, 0, bench.usable_cpus())
for name, flags in (("stream", 0), ("dense", ev.FLAG_DENSE_LAYOUT), ("short", ev.FLAG_COMPACT_LAYOUT)):
    t0 = time.perf_counter()
    with hip.DeviceBatch(batch, 0, flags) as d:
        t1 = time.perf_counter()
        d.genotype(sync=True)
        ms = min(d.genotype_timed(10) for _ in range(3)) / 10
        alg, res = d.bytes()
        print("%-7s create %.1f ms  pass %.4f ms  alg %.0f GB/s (%.2f of 8 TB/s)  resident %.0f MB  %.2f G units/s"
              % (name, (t1 - t0) * 1e3, ms, alg / ms / 1e6, alg / ms / 1e6 / 8000, res / 1e6, n / ms / 1e6), flush=True)
        if name == "stream":
            r_stream = d.results().rec.tobytes()
        elif name == "dense":
            print("   stream == dense results:", d.results().rec.tobytes() == r_stream)
