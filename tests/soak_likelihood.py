#!/usr/bin/env python
"""Differential soak: many random batches (random libraries, weights, flags; records and packed evidence) through the HIP path
and the C oracle; stops at the first difference.  Usage: tests/soak_likelihood.py [seconds [first_iteration]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repo root (this file lives in tests/)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_hip_parity as P
from oracle import c_oracle
from svtyper_amd import evidence as ev, hip, synth
import bench

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # seeds derive from the iteration number
t0, it, units, n_packed = time.time(), first, 0, 0
modes = {}            # table modes the hint-less / general-table resident batches ran in
fixture = bench.fixture_library()
while time.time() - t0 < budget:
    rng = np.random.default_rng(1000 + it)
    n_libs = int(rng.choice([1, 1, 2, 3, 5, 12]))
    libs = [fixture if (k == 0 and rng.random() < 0.5) else
            synth.normal_library(float(rng.uniform(150, 900)), float(rng.uniform(15, 260)), n=int(rng.integers(3000, 60000)),
                                 seed=int(rng.integers(1 << 30))) for k in range(n_libs)]
    kind = 6 if it % 13 == 7 else (it % 5 if it % 11 else 5)
    by_sample = 0       # > 0: the batch is site-major with this many samples per site (also checked sample-major)
    if kind == 0:
        b = P._fuzz_batch(5000 + it, libs, wide=bool(rng.integers(2)))
    elif kind == 1:
        b = synth.make_units(int(rng.integers(1, 30000)), 7000 + it, libs, svtype_mix=tuple(rng.dirichlet([2, 1, 1, 1])),
                             mean_frags=float(rng.uniform(5, 180)), sd_frags=float(rng.uniform(1, 60)), min_frags=0,
                             max_frags=int(rng.integers(60, 600)), frac_empty=0.02, frac_skip=0.01,
                             split_weight=float(rng.choice([1.0, 1.0, 0.5, 2.3])), disc_weight=float(rng.choice([1.0, 1.0, 0.25, 3.0])))
    elif kind == 2:
        b = synth.make_edge_cases(libs, seed=it)
    elif kind == 5:  # one library, enough units for two tiles per wave (launches of >= 221 184 units), ragged lengths
        libs = libs[:1]
        n_libs = 1
        # (every other time a launch of less than one round: 16 k ... 140 k units take the kernels with four / two lanes per unit)
        n5 = int(rng.integers(221_184, 300_000)) if rng.random() < 0.5 else int(rng.integers(16_000, 140_000))
        b = synth.make_units(n5, 3000 + it, libs, svtype_mix=tuple(rng.dirichlet([2, 1, 1, 1])),
                             mean_frags=float(rng.uniform(2, 25)), sd_frags=float(rng.uniform(1, 20)), min_frags=0,
                             max_frags=int(rng.integers(30, 260)), frac_empty=0.02, frac_skip=0.01)
    elif kind == 4:  # several samples with their own libraries: the streaming kernel's library windows (svt_unit.libs)
        by_sample = int(rng.choice([1, 2, 3, 8, 32, 40]))
        b = synth.make_multisample(int(rng.integers(1, 400)), by_sample, seed=it,
                                   mean_frags=float(rng.uniform(2, 80)), sd_frags=float(rng.uniform(1, 30)), min_frags=0,
                                   max_frags=int(rng.integers(40, 200)))
        if rng.random() < 0.5:      # units in any order, some dropped
            by_sample = 0
            b = synth.permute_units(b, rng.permutation(b.n_units)[:max(1, int(b.n_units * rng.uniform(0.3, 1.0)))])
    elif kind == 6:  # many samples and enough units for two tiles per wave in the library-window kernel
        by_sample = int(rng.choice([8, 32]))
        b = synth.make_multisample(int(rng.integers(221_184, 280_000)) // by_sample + 1, by_sample, seed=it,
                                   mean_frags=float(rng.uniform(2, 20)), sd_frags=float(rng.uniform(1, 12)), min_frags=0,
                                   max_frags=int(rng.integers(30, 120)))
    else:   # random bytes again, libraries close together per unit, MAPQs <= 127
        b = P._fuzz_batch(9000 + it, libs, wide=False)
        b.units["var_length"] = np.abs(b.units["var_length"])
        off = b.rec_offset.astype(np.int64)
        unit_of = np.repeat(np.arange(b.n_units), np.diff(off))
        base = rng.integers(0, max(1, n_libs - 3), b.n_units)
        lib = base[unit_of] + rng.integers(0, min(4, n_libs), b.n_records)
        fl = b.records["flags"] & ~np.uint32(0xff << ev.REC_LIB_SHIFT)
        b.records["flags"] = fl | (lib.astype(np.uint32) << ev.REC_LIB_SHIFT)
        if n_libs > 1:
            b.records["mapq_a"] &= 0x7f
            b.records["mapq_b"] &= 0x7f
    for flags in P.ALL_FLAGS:
        got = hip.genotype_batch(b, 0, flags)
        want = c_oracle.genotype_batch(b, flags & ev.FLAG_SSO_ASSOCIATION)
        try:
            P.assert_parity(got, want)
        except AssertionError as e:
            print("MISMATCH at iteration %d (kind %d, %d libs, flags %d): %s" % (it, kind, n_libs, flags, e))
            sys.exit(1)
    if n_libs > 1 or it % 3 == 0:   # resident batches with every table in L2 and, for several libraries, without hints (one window, or windows read off the records)
        nh = synth.permute_units(b, np.arange(b.n_units))
        nh.units["libs"] = 0
        for flags in (0, ev.FLAG_SSO_ASSOCIATION):
            ref_bytes = hip.genotype_batch(b, 0, flags).rec.tobytes()
            for x, fl in ((nh, flags), (nh, flags | ev.FLAG_GENERAL_TABLES), (b, flags | ev.FLAG_GENERAL_TABLES))[0 if n_libs > 1 else 2:]:
                with hip.DeviceBatch(x, 0, fl) as d:
                    d.genotype(sync=True)
                    modes[d.table_mode()] = modes.get(d.table_mode(), 0) + 1
                    if d.results().rec.tobytes() != ref_bytes:
                        print("TABLE-PATH MISMATCH at iteration %d (kind %d, %d libs, flags %#x, mode %d)" % (it, kind, n_libs, fl, d.table_mode()))
                        sys.exit(1)
                    if it % 7 == 0:   # the placement audition swaps the batch's result / record buffers: the bytes stay
                        d.tune_placement(3, 2)
                        d.genotype(sync=True)
                        if d.results().rec.tobytes() != ref_bytes:
                            print("TUNE-PLACEMENT MISMATCH at iteration %d (kind %d, %d libs, flags %#x)" % (it, kind, n_libs, fl))
                            sys.exit(1)
    if by_sample > 1:   # the same units handed over sample-major, result records written site-major (svt_batch_result_order)
        sm, _ = synth.to_sample_major(b, by_sample)
        for flags in (0, ev.FLAG_SSO_ASSOCIATION, ev.FLAG_RESULT96, ev.FLAG_RESULT96 | ev.FLAG_SSO_ASSOCIATION):
            with hip.DeviceBatch(sm, 0, flags) as d:
                d.result_order(by_sample)
                d.genotype(sync=True)
                got = d.results()
                if got.rec.tobytes() != hip.genotype_batch(b, 0, flags & ev.FLAG_SSO_ASSOCIATION).rec.tobytes():
                    print("RESULT-ORDER MISMATCH at iteration %d (kind %d, %d samples, flags %d)" % (it, kind, by_sample, flags))
                    sys.exit(1)
                # the same batch with its records handed over in pieces (svt_batch_create_segments): same bytes
                cuts = sorted(int(c) for c in rng.integers(0, sm.n_records + 1, int(rng.integers(0, 6))))
                edges = [0] + cuts + [sm.n_records]
                seg = ev.SegmentedBatch(sm.rec_offset, sm.units, [sm.records[a:b] for a, b in zip(edges[:-1], edges[1:])], sm.libs,
                                        sm.split_weight, sm.disc_weight)
                with hip.DeviceBatch.from_segments(seg, 0, flags) as ds:
                    ds.result_order(by_sample)
                    ds.genotype(sync=True)
                    if ds.results().rec.tobytes() != got.rec.tobytes():
                        print("SEGMENTS MISMATCH at iteration %d (kind %d, %d samples, flags %d, cuts %s)" % (it, kind, by_sample, flags, cuts))
                        sys.exit(1)
                # QUAL on the device (128-byte records site-major; tagged 96-byte records scattered by tag) = the host's running sum
                init = rng.uniform(-3.0, 40.0, b.n_units // by_sample) if rng.random() < 0.5 else None
                if d.site_qual(by_sample, init).tobytes() != np.asarray(hip.site_qual_host(got, by_sample, init)).tobytes():
                    print("SITE-QUAL MISMATCH at iteration %d (kind %d, %d samples, flags %d)" % (it, kind, by_sample, flags))
                    sys.exit(1)
    try:                # the packed evidence format, where it can hold the batch: same bytes as the canonical pass
        packed = hip.PackedEvidence(b)
    except hip.SvtyperHipError:
        packed = None
    if packed is not None:
        with packed:
            n_packed += 1
            for flags in (0, ev.FLAG_SSO_ASSOCIATION):
                ref_bytes = hip.genotype_batch(b, 0, flags).rec.tobytes()
                if hip.genotype_packed(packed, 0, flags).rec.tobytes() != ref_bytes:
                    print("PACKED MISMATCH at iteration %d (kind %d, %d libs, flags %d)" % (it, kind, n_libs, flags))
                    sys.exit(1)
                # the route that encodes ahead of the wire (ranges of whole units; small batches take the plain sequence)
                os.environ["SVT_PACK_RANGE_UNITS"] = str(int(rng.choice([256, 4096, 33000, 70000])))
                if hip.genotype_packed_from_records(b, 0, flags | (ev.FLAG_RESULT96 if it % 2 else 0)).rec.tobytes() != ref_bytes:
                    print("PACKED-FROM-RECORDS MISMATCH at iteration %d (kind %d, flags %d, ranges of %s units)" % (it, kind, flags, os.environ["SVT_PACK_RANGE_UNITS"]))
                    sys.exit(1)
    it += 1
    units += b.n_units
print("soak ok: iterations %d..%d, %d units, %d flag combinations each, %d batches also as packed evidence, table modes of the "
      "hint-less / general-table batches %s, %.0f s" % (first, it - 1, units, len(P.ALL_FLAGS), n_packed, sorted(modes.items()), time.time() - t0))
