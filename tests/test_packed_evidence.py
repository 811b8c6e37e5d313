"""Packed evidence (svt_packed_evidence): the host encoder svt_pack_evidence pinned on the CPU by an independent
decoder (oracle/py_packed.py) against the oracle on the canonical records, and -- on the GPU -- the pass over packed
evidence against the oracle and byte-for-byte against the pass over the canonical records."""
import numpy as np
import pytest

from svtyper_amd import evidence as ev
from svtyper_amd import synth


def _tallies_bits(x):
    return np.ascontiguousarray(x).view(np.uint64)


def _three_libraries(fixture_library):
    return [fixture_library, synth.normal_library(420.0, 95.0, seed=3), synth.normal_library(270.0, 40.0, seed=5)]


@pytest.mark.parametrize("n_libs", [1, 3])
@pytest.mark.parametrize("sso", [0, ev.FLAG_SSO_ASSOCIATION])
def test_encoder_against_an_independent_decoder(fixture_library, sso, n_libs):
    """CPU only: decode the slots the encoder wrote and redo the reference's arithmetic on them.  n_libs = 3: records of three
    libraries interleaved inside every unit (library switches in the pair stream, the small-deletion gate per library), and
    the configs[4] shape -- every sample's units with that sample's one to three libraries."""
    from oracle import c_oracle, py_packed
    from svtyper_amd import hip
    libs = _three_libraries(fixture_library)[:n_libs]
    batches = [synth.make_edge_cases(libs, seed=11).slice(0, 300),
               synth.make_units(400, 7, libs, svtype_mix=(0.5, 0.2, 0.2, 0.1), min_frags=0, mean_frags=25, sd_frags=20)]
    b = synth.make_units(300, 8, libs, svtype_mix=(0.6, 0.2, 0.1, 0.1), mean_frags=30, sd_frags=10)
    b.records["mapq_a"][::3] = 37                      # wide entries, every alignment case
    b.records["mapq_b"][::7] = 0
    b.units["var_length"][::5] = 3                      # below the small-deletion gate
    b.units["pos_delta"][::5] = 3
    batches.append(b)
    if n_libs > 1:
        batches.append(synth.make_multisample(40, 6, seed=3, mean_frags=40, sd_frags=20, min_frags=0, max_frags=120))
        g = synth.make_units(200, 9, libs, svtype_mix=(1.0, 0, 0, 0), mean_frags=30, sd_frags=10)
        g.units["pos_delta"][:] = 150                  # gated against library 0 (2 sd = 160) and 1 (190), not against library 2 (80)
        g.units["var_length"][:] = 150
        batches.append(g)
    for batch in batches:
        want = c_oracle.genotype_batch(batch, flags=sso).tallies
        with hip.PackedEvidence(batch) as p:
            assert p.n_units == batch.n_units and p.n_records == batch.n_records and p.c.n_libs == len(batch.libs)
            so = p.slot_offset()
            assert so[0] == 0 and so[-1] == p.c.n_slots and np.all(np.diff(so.astype(np.int64)) >= 0)
            got = py_packed.tally_packed(p.slots(), so, batch.units, list(batch.libs), int(p.c.common_mapq), bool(sso))
            if n_libs > 1:   # the switches are there, and only where a library really changes hands
                half = p.slots().view(np.uint16).reshape(-1)
                switches = int((((half & 0x8007) == 0) & (half != 0)).sum())
                assert 0 < switches <= batch.n_records
        skip = (batch.units["flags"] & ev.UNIT_SKIP) != 0       # (the oracle blanks skipped units; the decoder has no epilogue)
        assert np.array_equal(_tallies_bits(got[~skip]), _tallies_bits(want[~skip]))


def test_pack_rejects_what_the_format_cannot_hold(fixture_library):
    from svtyper_amd import hip
    wide = synth.make_units(50, 3, [synth.normal_library(3000.0, 900.0, seed=5)])
    assert hip.PackedEvidence.try_pack(wide) is None
    wide2 = synth.make_units(50, 3, [fixture_library, synth.normal_library(3000.0, 900.0, seed=5)])    # ... any of its libraries
    assert hip.PackedEvidence.try_pack(wide2) is None
    two = synth.make_units(50, 3, [fixture_library, synth.normal_library(420.0, 95.0, seed=3)])
    two.records["flags"][9] |= 2 << ev.REC_LIB_SHIFT       # a library the batch does not have
    with pytest.raises(hip.SvtyperHipError) as e:
        hip.PackedEvidence(two)
    assert "lib index" in str(e.value)
    neg = synth.make_units(50, 3, [fixture_library], svtype_mix=(1.0, 0, 0, 0))
    neg.units["var_length"][3] = -7
    assert hip.PackedEvidence.try_pack(neg) is None
    bad = synth.make_units(50, 3, [fixture_library])
    bad.records["flags"][5] |= 1 << 28
    with pytest.raises(hip.SvtyperHipError) as e:
        hip.PackedEvidence(bad)
    assert "reserved/undefined bits" in str(e.value)
    bad = synth.make_units(50, 3, [fixture_library])
    bad.records["flags"][7] = ev.REC_ALT_STRADDLE
    with pytest.raises(hip.SvtyperHipError):
        hip.PackedEvidence(bad)
    empty = synth.make_units(0, 3, [fixture_library])
    with hip.PackedEvidence(empty) as p:
        assert p.n_units == 0 and p.c.n_slots == 0
    ok = synth.make_units(2000, 3, [fixture_library])
    with hip.PackedEvidence(ok) as p:
        assert p.nbytes < 0.3 * (16 * ok.n_records)         # the point of the format


def _both(batch, flags):
    from oracle import c_oracle
    from svtyper_amd import hip
    with hip.PackedEvidence(batch) as p:
        got = hip.genotype_packed(p, flags=flags)
    canon = hip.genotype_batch(batch, flags=flags)
    assert got.rec.tobytes() == canon.rec.tobytes(), "packed pass differs from the pass over the canonical records"
    return got, c_oracle.genotype_batch(batch, flags=flags)


@pytest.mark.gpu
@pytest.mark.parametrize("sso", [0, ev.FLAG_SSO_ASSOCIATION])
def test_packed_pass_parity(hip_device, fixture_library, sso):
    from test_hip_parity import assert_parity, _sweep_batch
    batches = {
        "edge": synth.make_edge_cases([fixture_library], seed=11),
        "c2": synth.make_config("c2_del_100k", [fixture_library], n_units=20_000),
        "c3": synth.make_config("c3_mixed_1m", [fixture_library], n_units=30_000),
        "weights": synth.make_units(4000, 17, [fixture_library], svtype_mix=(0.6, 0.2, 0.1, 0.1), split_weight=0.7, disc_weight=1.9),
    }
    nb = len(fixture_library.hist)
    for svtype in (0, 1, 2):
        batches["sweep%d" % svtype] = _sweep_batch(fixture_library, [0, 1, 37, nb - 1, nb, nb + 1, 3 * nb, 100_000, 2**29], svtype)
    for n in (1, 63, 64, 65, 255, 257, 4097):
        batches["tiny%d" % n] = synth.make_units(n, n, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), min_frags=0,
                                                 mean_frags=30, sd_frags=30)
    # several libraries: interleaved inside the units, and per sample (the configs[4] shape, site-major and sample-major)
    libs = _three_libraries(fixture_library)
    batches["edge3"] = synth.make_edge_cases(libs, seed=11)
    batches["mixed3"] = synth.make_units(30_000, 23, libs, svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=40, sd_frags=30, min_frags=0,
                                         frac_empty=0.03, frac_skip=0.02)
    batches["c5"] = synth.make_multisample(600, 32, seed=13, mean_frags=40, sd_frags=20, min_frags=0, max_frags=150)
    batches["c5_by_sample"] = synth.to_sample_major(batches["c5"], 32)[0]
    for name, batch in batches.items():
        got, want = _both(batch, sso)
        assert_parity(got, want)
    want = batches["edge"]
    got, oracle = _both(want, sso)
    assert (oracle.gt == ev.GT_BLANK).any() and (oracle.gt == ev.GT_SKIPPED).any() and (oracle.gt == ev.GT_MISSING).any()


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", ["all_common", "all_wide", "alternate", "runs", "random", "mapq255", "vote_other"])
def test_packed_pair_entry_patterns(hip_device, fixture_library, pattern):
    """every mix of one-half-word and wide pair entries (and the no-op half-words that align the wide ones)"""
    import zlib
    from test_hip_parity import assert_parity
    rng = np.random.default_rng(zlib.crc32(pattern.encode()))
    batch = synth.make_units(3000, 61, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=40, sd_frags=25, min_frags=0)
    n = batch.n_records
    a, b = batch.records["mapq_a"], batch.records["mapq_b"]
    if pattern == "all_common":
        a[:], b[:] = 60, 60
    elif pattern == "all_wide":
        a[:] = rng.integers(1, 60, n); b[:] = 60 - a // 2
    elif pattern == "alternate":
        a[:], b[:] = 60, 60
        a[::2] = 37
    elif pattern == "runs":
        a[:], b[:] = 60, 60
        wide = (np.arange(n) % 11) < 3
        a[wide], b[wide] = 23, 59
    elif pattern == "random":
        a[:] = rng.choice([60, 60, 60, 0, 1, 40, 255], n); b[:] = rng.choice([60, 60, 60, 0, 13, 255], n)
    elif pattern == "mapq255":
        a[:], b[:] = 255, 255
        a[::5] = 128
    else:
        a[:], b[:] = 40, 13
        a[::4], b[::4] = 60, 60
    for flags in (0, ev.FLAG_SSO_ASSOCIATION):
        got, want = _both(batch, flags)
        assert_parity(got, want)


@pytest.mark.gpu
def test_packed_resident_batch_and_long_units(hip_device, fixture_library):
    """DeviceBatch.from_packed: re-runs are idempotent, bytes() reports the slots, results bind to a caller buffer;
    units with thousands of records (log10 table through L2, last sort bucket)."""
    from svtyper_amd import hip
    from test_hip_parity import assert_parity
    from oracle import c_oracle
    batch = synth.make_units(5000, 5, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1))
    with hip.PackedEvidence(batch) as p, hip.DeviceBatch.from_packed(p, hip_device) as d:
        assert d.layout_name() == "packed"
        alg, res = d.bytes()
        assert alg == batch.algorithmic_bytes() and res == 16 * int(p.c.n_slots) + 28 * batch.n_units
        d.genotype(sync=True)
        first = d.results()
        d.genotype(sync=True)
        assert d.results().rec.tobytes() == first.rec.tobytes()
    assert_parity(first, c_oracle.genotype_batch(batch))
    counts = np.array([3000, 5, 7000, 0, 1, 2600])
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    r = np.zeros(int(off[-1]), ev.RECORD_DTYPE)
    r["flags"] = np.uint32(ev.REC_ALT_STRADDLE | ev.REC_HAS_PAIR)
    r["mapq_a"], r["mapq_b"], r["rs_a"], r["ospan_len"] = 60, 60, 60, 100_000
    u = np.zeros(len(counts), ev.UNIT_DTYPE)
    u["var_length"], u["pos_delta"] = 5000, 5000
    long_units = ev.EvidenceBatch(off, u, r, [fixture_library], 1.0, 1.0)
    got, want = _both(long_units, 0)
    assert_parity(got, want)


def test_vector_and_scalar_encoders_write_the_same_slots(fixture_library):
    """svt_pack.cpp encodes sixteen records at a time with AVX-512 where the host has it (runs of one-half-word pair
    entries by compressing stores, wide entries / weight entries from mask bits, groups with a continuation record one
    record at a time) and record by record otherwise (SVT_PACK_SCALAR=1 forces that form): same slots, same offsets,
    on batches that mix every kind of record, unit lengths around the group size, gated deletions and empty units."""
    import os
    from svtyper_amd import hip
    rng = np.random.default_rng(2026)
    for trial in range(11):
        # (trials 6-8: three libraries -- interleaved record by record, then in runs of a dozen and of forty records; 9: forty
        # libraries, a unit's records all over them (more than the vector form's window of sixteen: record by record); 10: the
        # configs[4] shape without the units' window hints (the window moves to each sample's libraries))
        libs = [fixture_library] if trial < 6 else _three_libraries(fixture_library)
        if trial == 9:
            libs = [synth.normal_library(250.0 + 7 * k, 40.0 + k, n=50_000, seed=k) for k in range(40)]
        mean = (5, 17, 33, 64, 100, 180, 33, 100, 180, 64, 50)[trial]
        if trial == 10:
            batch = synth.make_multisample(120, 24, seed=41, mean_frags=50, sd_frags=30, min_frags=0, max_frags=200)
            batch.units["libs"] = 0
        else:
            batch = synth.make_units(3000, 100 + trial, libs, svtype_mix=(0.6, 0.2, 0.1, 0.1), mean_frags=mean,
                                     sd_frags=(4, 9, 16, 20, 30, 60, 16, 30, 60, 20, 0)[trial], min_frags=0)
        if trial in (7, 8):
            run = (np.arange(batch.n_records) // (12 if trial == 7 else 40)) % 3
            batch.records["flags"] = (batch.records["flags"] & ~np.uint32(0xff00)) | (run.astype(np.uint32) << np.uint32(ev.REC_LIB_SHIFT))
        rec = batch.records
        n = batch.n_records
        firsts = set(int(x) for x in batch.rec_offset[:-1])
        # continuation records (never a unit's first), odd MAPQ pairs (wide pair entries), reference reads and candidates
        cont = np.array([i for i in rng.choice(n, n // 40, replace=False) if int(i) not in firsts], dtype=np.int64)
        rec["flags"][cont] = (rec["flags"][cont] & 0xff00) | ev.REC_CONTINUATION
        for name in ("ospan_len", "mapq_a", "mapq_b"):
            rec[name][cont] = 0
        odd = rng.choice(n, n // 5, replace=False)
        rec["mapq_a"][odd] = rng.integers(0, 256, odd.size)
        rec["mapq_b"][odd] = rng.integers(0, 256, odd.size)
        for name, share in (("rs_a", 3), ("rs_b", 3), ("seq_l", 9), ("seq_r", 9), ("clip_l", 11), ("clip_r", 11)):
            pick = rng.choice(n, n // share, replace=False)
            rec[name][pick] = rng.integers(0, 256, pick.size)
        rec["ospan_len"][rng.choice(n, n // 7, replace=False)] = rng.integers(0, 200000, n // 7)
        got = {}
        for mode in ("vector", "scalar"):
            if mode == "scalar":
                os.environ["SVT_PACK_SCALAR"] = "1"
            try:
                with hip.PackedEvidence(batch) as p:
                    got[mode] = (p.slots().tobytes(), p.slot_offset().tobytes(), int(p.c.common_mapq))
            finally:
                os.environ.pop("SVT_PACK_SCALAR", None)
        assert got["vector"] == got["scalar"], trial


def test_worker_count_does_not_change_the_slots(fixture_library):
    """The encoder's workers claim chunks of 256 units from a counter, so which worker encodes which chunk differs from
    call to call; the output is placed by chunk, not by worker: one, three, eight and forty workers (SVT_PACK_THREADS)
    must write the same slots and offsets -- also when there are fewer chunks than workers."""
    import os
    from svtyper_amd import hip
    for n_units in (100, 5000, 5001):
        libs = [fixture_library] if n_units != 5001 else _three_libraries(fixture_library)
        batch = synth.make_units(n_units, 77, libs, svtype_mix=(0.6, 0.2, 0.1, 0.1), mean_frags=40, sd_frags=30, min_frags=0)
        got = []
        keep = os.environ.get("SVT_PACK_THREADS")
        try:
            for nt in (1, 3, 8, 40):
                os.environ["SVT_PACK_THREADS"] = str(nt)
                with hip.PackedEvidence(batch) as p:
                    got.append((p.slots().tobytes(), p.slot_offset().tobytes(), int(p.c.common_mapq)))
        finally:
            if keep is None:
                os.environ.pop("SVT_PACK_THREADS", None)
            else:
                os.environ["SVT_PACK_THREADS"] = keep
        assert all(g == got[0] for g in got[1:])


def test_ranged_encoder_writes_the_same_arrays(fixture_library, monkeypatch):
    """host only: the encoder taken through ranges of whole units with a hand-over after each (what
    svt_genotype_packed_from_records drives, SVT_PACK_TEST_RANGES) leaves the arrays of the plain call -- ranges in order,
    every unit handed over once, offsets final at the hand-over."""
    from svtyper_amd import hip
    batch = synth.make_units(40_000, 11, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=25, sd_frags=20, min_frags=0,
                             frac_empty=0.03, frac_skip=0.02)
    with hip.PackedEvidence(batch) as plain:
        want = (plain.slots().tobytes(), plain.slot_offset().tobytes(), plain.nbytes)
    for ranges in ("256", "1000", "7000", "40000", "1000000"):
        monkeypatch.setenv("SVT_PACK_TEST_RANGES", ranges)
        with hip.PackedEvidence(batch) as p:
            assert (p.slots().tobytes(), p.slot_offset().tobytes(), p.nbytes) == want, ranges
    monkeypatch.setenv("SVT_PACK_TEST_RANGES", "512")
    monkeypatch.setenv("SVT_PACK_THREADS", "5")
    with hip.PackedEvidence(batch) as p:
        assert p.slots().tobytes() == want[0]
    bad = synth.make_units(9000, 3, [fixture_library])
    bad.records["flags"][bad.n_records - 5] |= 1 << 28     # an undefined flag bit in the LAST range: reported, nothing leaks
    with pytest.raises(hip.SvtyperHipError) as e:
        hip.PackedEvidence(bad)
    assert "reserved/undefined bits" in str(e.value)
    monkeypatch.delenv("SVT_PACK_TEST_RANGES")
    monkeypatch.delenv("SVT_PACK_THREADS")
    multi = synth.make_multisample(900, 16, seed=5, mean_frags=25, sd_frags=15, min_frags=0, max_frags=90)     # several libraries
    with hip.PackedEvidence(multi) as plain:
        want = (plain.slots().tobytes(), plain.slot_offset().tobytes(), plain.nbytes)
    for ranges in ("256", "3000"):
        monkeypatch.setenv("SVT_PACK_TEST_RANGES", ranges)
        with hip.PackedEvidence(multi) as p:
            assert (p.slots().tobytes(), p.slot_offset().tobytes(), p.nbytes) == want, ranges


@pytest.mark.gpu
def test_from_records_route_overlaps_and_equals_the_canonical_pass(hip_device, fixture_library, monkeypatch):
    """svt_genotype_packed_from_records (encode || upload || pass || download by unit ranges) returns the bytes of svt_genotype over the
    same records: both associations, 96-byte device records, page-locked and pageable output, many small ranges, the serial
    fallbacks (a small batch; more slots than estimated), its error paths, and batches of several libraries."""
    from svtyper_amd import hip
    batch = synth.make_units(150_000, 19, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=30, sd_frags=20, min_frags=0,
                             frac_empty=0.03, frac_skip=0.02)
    pinned = hip.pinned_results(batch.n_units)
    for flags in (0, ev.FLAG_SSO_ASSOCIATION, ev.FLAG_RESULT96, ev.FLAG_SSO_ASSOCIATION | ev.FLAG_RESULT96):
        want = hip.genotype_batch(batch, hip_device, flags & ev.FLAG_SSO_ASSOCIATION).rec.tobytes()
        assert hip.genotype_packed_from_records(batch, hip_device, flags).rec.tobytes() == want
        assert hip.genotype_packed_from_records(batch, hip_device, flags, out=pinned).rec.tobytes() == want
    want = hip.genotype_batch(batch, hip_device, 0).rec.tobytes()
    monkeypatch.setenv("SVT_PACK_RANGE_UNITS", "4096")          # ~37 ranges
    assert hip.genotype_packed_from_records(batch, hip_device, 0, out=pinned).rec.tobytes() == want
    monkeypatch.delenv("SVT_PACK_RANGE_UNITS")
    monkeypatch.setenv("SVT_PACKED_SERIAL", "1")                # the plain sequence
    assert hip.genotype_packed_from_records(batch, hip_device, 0).rec.tobytes() == want
    monkeypatch.delenv("SVT_PACKED_SERIAL")
    small = batch.slice(0, 5000)                                # below the pipeline's minimum: the plain sequence
    assert hip.genotype_packed_from_records(small, hip_device, 0).rec.tobytes() == hip.genotype_batch(small, hip_device, 0).rec.tobytes()
    assert hip.genotype_packed_from_records(batch.slice(0, 0), hip_device, 0).n_units == 0
    # records that pack badly (every pair entry wide, every record with candidates): more slots than the estimate -> the plain route
    dense = synth.make_units(40_000, 5, [fixture_library], mean_frags=30, sd_frags=5, min_frags=10)
    rng = np.random.default_rng(1)
    for f in ("mapq_a", "mapq_b", "rs_a", "rs_b", "seq_l", "seq_r", "clip_l", "clip_r"):
        dense.records[f] = rng.integers(1, 60, dense.n_records).astype(np.uint8)
    assert hip.genotype_packed_from_records(dense, hip_device, 0).rec.tobytes() == hip.genotype_batch(dense, hip_device, 0).rec.tobytes()
    # a contract violation in a late range is an error of the call; several libraries are not packable
    bad = synth.make_units(60_000, 3, [fixture_library])
    bad.records["flags"][bad.n_records - 5] |= 1 << 28
    with pytest.raises(hip.SvtyperHipError) as e:
        hip.genotype_packed_from_records(bad, hip_device, 0)
    assert "reserved/undefined bits" in str(e.value)
    # the device is still fine afterwards
    assert hip.genotype_packed_from_records(batch, hip_device, 0).rec.tobytes() == want
    # several libraries: interleaved inside the units, and the configs[4] shape sample-major (library switches; tables through L2)
    two = synth.make_units(60_000, 3, [fixture_library, synth.normal_library(400.0, 60.0, seed=2)], svtype_mix=(0.5, 0.2, 0.2, 0.1),
                           mean_frags=30, sd_frags=20, min_frags=0)
    c5 = synth.to_sample_major(synth.make_multisample(2500, 32, seed=7, mean_frags=30, sd_frags=15, min_frags=0, max_frags=120), 32)[0]
    for multi in (two, c5):
        for flags in (0, ev.FLAG_SSO_ASSOCIATION | ev.FLAG_RESULT96):
            want_m = hip.genotype_batch(multi, hip_device, flags & ev.FLAG_SSO_ASSOCIATION).rec.tobytes()
            assert hip.genotype_packed_from_records(multi, hip_device, flags).rec.tobytes() == want_m
            monkeypatch.setenv("SVT_PACK_RANGE_UNITS", "4096")
            assert hip.genotype_packed_from_records(multi, hip_device, flags).rec.tobytes() == want_m
            monkeypatch.delenv("SVT_PACK_RANGE_UNITS")
