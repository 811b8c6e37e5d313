#!/usr/bin/env python
"""Which kernel for a launch of less than one round?  The same resident batch through the streaming kernel (one tile per
wave), the cooperative kernel and the kernels with two / four lanes per unit (svt_debug_small_kind), byte-compared and timed,
by units and by records per unit.
    python tools/small_kinds.py [sso]"""
import ctypes as C
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from svtyper_amd import hip, synth, evidence as ev

lib = bench.fixture_library()
L = hip.load()
L.svt_debug_small_kind.argtypes = [C.c_int]
L.svt_debug_small_kind.restype = C.c_int
flags = ev.FLAG_RESULT96 | (ev.FLAG_SSO_ASSOCIATION if "sso" in sys.argv[1:] else 0)
KINDS = (("stream", 1), ("coop", 2), ("split2", 3), ("split4", 4))
if "c5" in sys.argv[1:]:
    # the configs[4] shape (sites x 32 samples, per-sample library windows, sample-major units -> site-major records): cooperative kernel n/a
    for n in (4_000, 16_000, 32_000, 64_000, 100_000, 131_000, 180_000):
        b = synth.make_multisample(max(1, n // 32), 32, seed=9, layout="sample")
        row, digests = [], set()
        for name, kind in (("stream", 1), ("split2", 3), ("split4", 4)):
            if kind == 3 and (flags & ev.FLAG_SSO_ASSOCIATION):
                continue
            L.svt_debug_small_kind(kind)
            with hip.DeviceBatch(b, 0, flags) as d:
                d.result_order(32)
                d.genotype(sync=True)
                digests.add(hashlib.sha1(d.results().rec.tobytes()).hexdigest()[:10])
                for _ in range(3):
                    d.genotype_timed(40)
                ms = min(d.genotype_timed(40) for _ in range(5)) / 40
            row.append("%s %.4f" % (name, ms))
        L.svt_debug_small_kind(0)
        print("c5 n %7d: %s  %s" % (b.n_units, " | ".join(row), "same bytes" if len(digests) == 1 else "DIFFERENT BYTES %s" % digests), flush=True)
    sys.exit(0)
for mean in (100, 400):
    for n in (2_000, 5_000, 10_000, 20_000, 30_000, 45_000, 65_000, 90_000, 131_000, 160_000):
        if mean == 400 and n > 65_000:
            continue
        b = synth.make_units(n, 300 + mean, [lib], svtype_mix=(0.7, 0.15, 0.15, 0.0), mean_frags=mean, sd_frags=mean // 4,
                             min_frags=mean // 5, max_frags=mean * 2)
        row, digests = [], set()
        for name, kind in KINDS:
            L.svt_debug_small_kind(kind)
            with hip.DeviceBatch(b, 0, flags) as d:
                d.genotype(sync=True)
                digests.add(hashlib.sha1(d.results().rec.tobytes()).hexdigest()[:10])
                for _ in range(3):
                    d.genotype_timed(40)
                ms = min(d.genotype_timed(40) for _ in range(5)) / 40
            row.append("%s %.4f" % (name, ms))
        L.svt_debug_small_kind(0)
        print("F=%3d n %7d: %s  %s" % (mean, n, " | ".join(row), "same bytes" if len(digests) == 1 else "DIFFERENT BYTES %s" % digests), flush=True)
