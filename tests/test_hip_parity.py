"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bars (BASELINE.json north_star): integer outputs (GT, GQ, QR, QA, DP, RO, AO, RS, AS, ASC, RP,
AP) bit-exact; GL and SQ within 1e-6 absolute.  The tallies and GL are in fact produced by the
same sequence of binary64 operations on both sides, so they are additionally required to be
bit-identical; only SQ goes through the device's pow/log.
"""
import numpy as np
import pytest

from svtyper_amd import evidence as ev
from svtyper_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-6  # north_star tolerance on GL / SQ


def assert_parity(got: ev.Results, want: ev.Results, exact_float=True):
    assert np.array_equal(got.gt, want.gt), "GT mismatch at %s" % np.nonzero(got.gt != want.gt)[0][:10]
    for i, name in enumerate(ev.COUNT_NAMES):
        bad = np.nonzero(got.counts[:, i] != want.counts[:, i])[0]
        assert bad.size == 0, "%s mismatch at units %s: %s vs %s" % (
            name, bad[:10], got.counts[bad[:10], i], want.counts[bad[:10], i])
    assert np.max(np.abs(got.gl - want.gl), initial=0.0) <= TOL
    assert np.max(np.abs(got.sq - want.sq), initial=0.0) <= TOL
    assert np.max(np.abs(got.tallies - want.tallies), initial=0.0) <= TOL
    if exact_float:
        assert np.array_equal(np.ascontiguousarray(got.tallies).view(np.uint64),
                              np.ascontiguousarray(want.tallies).view(np.uint64)), "tallies not bit-identical"
        assert np.array_equal(np.ascontiguousarray(got.gl).view(np.uint64),
                              np.ascontiguousarray(want.gl).view(np.uint64)), "GL not bit-identical"
    assert not got.rec["pad"].any()


def run_both(batch, flags=0, device=0):
    """HIP (canonical records through svt_genotype) and the oracle; where the packed format can hold the batch the
    same units also go through svt_pack_evidence -> svt_genotype_packed and must come back with the same bytes."""
    from oracle import c_oracle
    from svtyper_amd import hip
    got = hip.genotype_batch(batch, device=device, flags=flags)
    want = c_oracle.genotype_batch(batch, flags=flags & ev.FLAG_SSO_ASSOCIATION)
    packed = hip.PackedEvidence.try_pack(batch)
    if packed is not None:
        with packed:
            again = hip.genotype_packed(packed, device=device, flags=flags)
        assert again.rec.tobytes() == got.rec.tobytes(), "packed pass differs from the pass over the canonical records"
    return got, want


ALL_FLAGS = [0, ev.FLAG_SSO_ASSOCIATION, ev.FLAG_RESULT96, ev.FLAG_SSO_ASSOCIATION | ev.FLAG_RESULT96]   # (96-byte device records: same host results)


def oracle_flags(flags):
    return flags & ev.FLAG_SSO_ASSOCIATION


@pytest.mark.parametrize("flags", ALL_FLAGS)
def test_edge_cases(hip_device, fixture_library, flags):
    batch = synth.make_edge_cases([fixture_library], seed=11)
    got, want = run_both(batch, flags)
    assert_parity(got, want)
    # the edge batch must really exercise the special outcomes
    assert (want.gt == ev.GT_BLANK).any() and (want.gt == ev.GT_SKIPPED).any()
    assert (want.gt == ev.GT_MISSING).any(), "no underflow (GT './.') case generated"
    for g in (0, 1, 2):
        assert (want.gt == g).any()


@pytest.mark.parametrize("flags", ALL_FLAGS)
def test_c2_slice(hip_device, fixture_library, flags):
    """BASELINE.json configs[1] (100k DEL sites, 1 library), a 20k-unit slice."""
    batch = synth.make_config("c2_del_100k", [fixture_library], n_units=20_000)
    got, want = run_both(batch, flags)
    assert_parity(got, want)


def test_c2_at_its_full_size(hip_device, fixture_library):
    """BASELINE.json configs[1] literally: all 100 000 DEL sites (10 M records) against the oracle, classic association, both
    device record forms (0.2 s of GPU; the C oracle takes a second)."""
    from oracle import c_oracle
    from svtyper_amd import hip
    batch = synth.make_config("c2_del_100k", [fixture_library])
    assert batch.n_units == 100_000 and (batch.units["svtype"] == 0).all()
    want = c_oracle.genotype_batch(batch, flags=0)
    for flags in (0, ev.FLAG_RESULT96):
        assert_parity(hip.genotype_batch(batch, device=hip_device, flags=flags), want)
        with hip.DeviceBatch(batch, hip_device, flags) as d:
            d.genotype(sync=True)
            assert_parity(d.results(), want)


@pytest.mark.parametrize("flags", ALL_FLAGS)
def test_c3_slice_mixed(hip_device, fixture_library, flags):
    """configs[2]: mixed DEL/DUP/INV."""
    batch = synth.make_config("c3_mixed_1m", [fixture_library], n_units=30_000)
    got, want = run_both(batch, flags)
    assert_parity(got, want)
    assert len(np.unique(batch.units["svtype"])) == 3


@pytest.mark.parametrize("flags", ALL_FLAGS)
def test_multi_library(hip_device, fixture_library, flags):
    libs = [fixture_library, synth.normal_library(420.0, 95.0, seed=3), synth.normal_library(280.0, 40.0, seed=4)]
    batch = synth.make_units(5000, 99, libs, svtype_mix=(0.5, 0.2, 0.2, 0.1))
    got, want = run_both(batch, flags)
    assert_parity(got, want)


def test_wide_geometry_general_mode(hip_device, fixture_library):
    """DEL lengths beyond 2^30 and histogram keys far from 0 force the 64-bit 'general' kernel."""
    lib = synth.normal_library(300.0, 50.0, seed=6)
    batch = synth.make_units(3000, 8, [lib, fixture_library], svtype_mix=(0.6, 0.2, 0.1, 0.1))
    batch.units["var_length"][::7] = 2**30 + 12345
    batch.units["pos_delta"][::7] = 2**30 + 12346
    big = batch.units["var_length"][batch.units["svtype"] == 0].max()
    batch.records["ospan_len"][::5] = np.minimum(2**31 - 1, batch.records["ospan_len"][::5].astype(np.int64) + big)
    for flags in ALL_FLAGS:
        got, want = run_both(batch, flags)
        assert_parity(got, want)


def test_integral_nondel_var_length(hip_device):
    """mean + 3 sd integral: the float Counter key of parsers.py:874-878 matches integer bins."""
    lib = synth.normal_library(300.0, 50.0, seed=5)
    lib.mean, lib.sd = 300.0, 50.0  # v = 450.0 exactly
    batch = synth.make_units(4000, 5, [lib], svtype_mix=(0.0, 0.4, 0.4, 0.2))
    got, want = run_both(batch)
    assert_parity(got, want)


def test_two_tiles_per_wave(hip_device, fixture_library):
    """From 221 184 units per launch on, a one-library pass gives every wave two 64-unit tiles in snake order
    (a workgroup sorts 512 units): ragged lengths incl. empty units, a last workgroup that is not full, both associations."""
    n = 221_184 + 512 * 3 + 77
    batch = synth.make_units(n, 29, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=14, sd_frags=12,
                             min_frags=0, max_frags=150, frac_empty=0.03, frac_skip=0.01)
    for flags in ALL_FLAGS:
        got, want = run_both(batch, flags)
        assert_parity(got, want)
    # the same units in launches too small for two tiles give the same bytes
    from svtyper_amd import hip
    lo = hip.genotype_batch(batch.slice(0, 100_000), device=hip_device)
    assert np.array_equal(lo.rec, hip.genotype_batch(batch, device=hip_device).rec[:100_000])
    # a wider histogram leaves no room for the log10 table beside the other tables: the epilogue of a wave's first
    # tile then borrows the ring (LDS-DMA copy of the table) that its second tile streams through right after
    broad = synth.normal_library(1500.0, 400.0, seed=9)
    assert len(broad.hist) > 2500
    wide = synth.make_units(n, 31, [broad], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=14, sd_frags=12, min_frags=0, max_frags=150)
    got, want = run_both(wide, 0)
    assert_parity(got, want)


def test_sso_rare_continuations(hip_device, fixture_library):
    """singlesample association: a block of records without any continuation record takes the select-free form of
    the fragment-local sums, a block with one the general form -- here both kinds alternate inside every unit."""
    batch = synth.make_units(6000, 23, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=90, sd_frags=40, min_frags=0)
    rng = np.random.default_rng(23)
    for rate in (0.002, 0.05):
        b = synth.permute_units(batch, np.arange(batch.n_units))
        cont = rng.random(b.n_records) < rate
        b.records["flags"][cont] |= np.uint32(ev.REC_CONTINUATION)
        assert cont.any() and not cont.all()
        for flags in ALL_FLAGS:
            got, want = run_both(b, flags)
            assert_parity(got, want)


def test_weights(hip_device, fixture_library):
    batch = synth.make_units(4000, 17, [fixture_library], svtype_mix=(0.6, 0.2, 0.1, 0.1),
                             split_weight=0.7, disc_weight=1.9)
    got, want = run_both(batch)
    assert_parity(got, want)


def test_empty_and_tiny_batches(hip_device, fixture_library):
    from svtyper_amd import hip
    b0 = synth.make_units(0, 1, [fixture_library])
    r0 = hip.genotype_batch(b0)
    assert r0.n_units == 0
    for n in (1, 63, 64, 65, 4095, 4097):
        b = synth.make_units(n, n, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), min_frags=0,
                             mean_frags=30, sd_frags=30)
        got, want = run_both(b)
        assert_parity(got, want)


def test_invalid_records_rejected(hip_device, fixture_library):
    from svtyper_amd import hip
    b = synth.make_units(100, 3, [fixture_library])
    b.records["flags"][5] |= 7 << ev.REC_LIB_SHIFT  # only one library
    with pytest.raises(hip.SvtyperHipError):
        hip.genotype_batch(b)
    b = synth.make_units(100, 3, [fixture_library])
    b.records["flags"][7] = ev.REC_ALT_STRADDLE  # straddle bit without HAS_PAIR
    with pytest.raises(hip.SvtyperHipError):
        hip.genotype_batch(b)
    b = synth.make_units(100, 3, [fixture_library])
    b.records["flags"][9] |= 1 << 28  # undefined flag bit
    with pytest.raises(hip.SvtyperHipError):
        hip.genotype_batch(b)


def test_bayes_grid_via_kernel(hip_device, fixture_library):
    """Drive (QR, QA, is_dup) through the kernel with MAPQ-255 evidence (weight exactly 1.0):
    QR ref-seq reads and QA alt pairs per unit, grid 0..330 x 0..330 x {DEL-like, DUP}."""
    from oracle import c_oracle
    qs = np.array([0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 100, 144, 200, 233, 300, 330])
    units, recs, offs = [], [], [0]
    for svt in (ev.SVTYPE_CODE["INV"], ev.SVTYPE_CODE["DUP"]):
        for qr in qs:
            for qa in qs:
                n = max(qr, qa)
                r = np.zeros(n, ev.RECORD_DTYPE)
                r["mapq_a"] = 255
                r["mapq_b"] = 255
                fl = np.full(n, ev.REC_HAS_PAIR, np.uint32)
                r["rs_a"][:qr] = 255
                fl[:qa] |= ev.REC_ALT_STRADDLE
                r["flags"] = fl
                recs.append(r)
                offs.append(offs[-1] + n)
                u = np.zeros(1, ev.UNIT_DTYPE)
                u["svtype"] = svt
                units.append(u)
    batch = ev.EvidenceBatch(np.array(offs, np.uint64), np.concatenate(units), np.concatenate(recs),
                             [fixture_library])
    got, want = run_both(batch)
    assert_parity(got, want)
    # spot-check the pinned reference values (SURVEY.md 8c iii)
    k = 0
    for svt in ("INV", "DUP"):
        for qr in qs:
            for qa in qs:
                if (qr, qa) == (0, 0):
                    assert got.gt[k] == ev.GT_BLANK
                elif (qr, qa) == (0, 1) and svt == "INV":
                    # alt_span >= 1 and no splitters: zeroing rule keeps QA = 1
                    assert got.gt[k] == 2 and got.count("GQ")[k] == 2
                    assert abs(got.sq[k] - 31.464381352857743) < TOL
                k += 1


def test_multisample_per_sample_libraries(hip_device):
    """BASELINE.json configs[4] shape at test size: 150 sites x 32 samples, 1..3 libraries per sample
    (too many tables for LDS -> the general kernel), both layouts and associations."""
    batch = synth.make_multisample(150, 32, seed=5, mean_frags=40, sd_frags=15, min_frags=5, max_frags=90)
    assert len(batch.libs) >= 32 and batch.n_units == 150 * 32
    assert (batch.units["sample"][:32] == np.arange(32)).all()
    for flags in ALL_FLAGS:
        got, want = run_both(batch, flags)
        assert_parity(got, want)


def _fuzz_batch(seed, libs, wide):
    """Uniformly random bytes inside the record contract: any MAPQ 0..255 anywhere, any combination of the
    defined flag bits, spans around (and far from) the histogram, arbitrary svtype / lengths / weights."""
    rng = np.random.default_rng(seed)
    n = 3000
    F = rng.integers(0, 220, n)
    F[rng.random(n) < 0.05] = 0
    off = np.zeros(n + 1, np.uint64)
    np.cumsum(F, out=off[1:])
    R = int(off[-1])
    rec = np.zeros(R, ev.RECORD_DTYPE)
    for fld in ("mapq_a", "mapq_b", "rs_a", "rs_b", "seq_l", "seq_r", "clip_l", "clip_r"):
        common = rng.choice([0, 60, 255, 10, 20, 3], R)
        rec[fld] = np.where(rng.random(R) < 0.4, 0, np.where(rng.random(R) < 0.5, common, rng.integers(0, 256, R)))
    pair = rng.random(R) < 0.8
    bits = rng.integers(0, 8, R) * pair                      # straddle bits only with HAS_PAIR
    cont = (rng.random(R) < 0.1).astype(np.uint32) * ev.REC_CONTINUATION
    lib = rng.integers(0, len(libs), R).astype(np.uint32)
    rec["flags"] = bits.astype(np.uint32) | cont | (pair.astype(np.uint32) * ev.REC_HAS_PAIR) | (lib << ev.REC_LIB_SHIFT)
    span = rng.integers(0, 1500, R)
    span = np.where(rng.random(R) < 0.1, rng.integers(0, 2**31 - 1, R), span)
    rec["ospan_len"] = span
    units = np.zeros(n, ev.UNIT_DTYPE)
    units["svtype"] = rng.integers(0, 4, n)
    vl = rng.integers(-50, 1500, n)
    if wide:
        vl = np.where(rng.random(n) < 0.2, rng.integers(-2**31, 2**31 - 1, n), vl)
    units["var_length"] = vl
    units["pos_delta"] = np.where(rng.random(n) < 0.5, rng.integers(-10, 400, n), rng.integers(-2**31, 2**31 - 1, n))
    units["sample"] = rng.integers(0, 65536, n)
    units["flags"] = (rng.random(n) < 0.02) * ev.UNIT_SKIP
    sw, dw = rng.choice([1.0, 0.5, 2.0, 0.0, 1.3]), rng.choice([1.0, 0.25, 3.0, 0.0, 0.9])
    return ev.EvidenceBatch(off, units, rec, libs, float(sw), float(dw))


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_fuzz_random_records(hip_device, fixture_library, seed):
    libs_sets = {
        1: [fixture_library],
        2: [fixture_library, synth.normal_library(420.0, 95.0, n=50000, seed=3)],
        3: [synth.normal_library(300.0, 50.0, n=30000, seed=9)],
        4: [synth.normal_library(250.0 + 30 * k, 40.0 + 5 * k, n=20000, seed=20 + k) for k in range(40)],
    }
    libs = libs_sets[seed]
    if seed == 3:
        libs[0].mean, libs[0].sd = 300.0, 50.0          # integral mean + 3 sd -> exact float-key kernel
    batch = _fuzz_batch(100 + seed, libs, wide=(seed % 2 == 0))
    for flags in ALL_FLAGS:
        got, want = run_both(batch, flags)
        assert_parity(got, want)


# ------------------------------------------------------------------------------------------
# BASELINE.json's full size (configs[2]: 1 M units, ~100 M records) through size-independent
# properties -- the oracle only sees a bounded random sample of it
# ------------------------------------------------------------------------------------------
def _digest(res: ev.Results) -> np.ndarray:
    """per-unit checksum over the whole 128-byte result record"""
    words = np.ascontiguousarray(res.rec).view(np.uint64).reshape(len(res.rec), -1)
    mult = (np.arange(words.shape[1], dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1))
    with np.errstate(over="ignore"):
        return (words * mult).sum(axis=1, dtype=np.uint64)


@pytest.fixture(scope="module")
def full_c3(fixture_library):
    import multiprocessing as mp
    n, chunk = synth.CONFIGS["c3_mixed_1m"]["n_units"], 62_500
    jobs = [(chunk, synth.BASE_SEED + 3 + 1000 * i, [fixture_library], synth.CONFIGS["c3_mixed_1m"]["svtype_mix"])
            for i in range(n // chunk)]
    with mp.get_context("fork").Pool(min(16, len(jobs))) as pool:
        parts = pool.starmap(_c3_chunk, jobs)
    return ev.concat_batches(parts)


def _c3_chunk(n, seed, libs, mix):
    return synth.make_units(n, seed, libs, svtype_mix=mix)


def test_full_size_properties(hip_device, full_c3):
    from oracle import c_oracle
    from svtyper_amd import hip
    batch = full_c3
    n = batch.n_units
    assert n == 1_000_000 and batch.n_records > 90_000_000
    with hip.DeviceBatch(batch, device=hip_device) as d:
        d.genotype(sync=True)
        first = d.results()
        d.genotype(sync=True)                       # idempotence: a second pass over the resident batch
        again = d.results()
    assert np.array_equal(first.rec, again.rec)
    base = _digest(first)
    # every result record is fully written and the padding stays zero
    assert not first.rec["pad"].any()
    assert np.isin(first.gt, (0, 1, 2, ev.GT_MISSING, ev.GT_BLANK)).all()

    # split invariance: a unit's result does not depend on which other units share its batch / tile
    cut = 333_337
    lo = hip.genotype_batch(batch.slice(0, cut), device=hip_device)
    hi = hip.genotype_batch(batch.slice(cut, n), device=hip_device)
    assert np.array_equal(np.concatenate([_digest(lo), _digest(hi)]), base)

    # permutation invariance (the library re-sorts units by length; results come back in input order)
    rng = np.random.default_rng(99)
    order = rng.permutation(n)
    perm = hip.genotype_batch(synth.permute_units(batch, order), device=hip_device)
    assert np.array_equal(_digest(perm), base[order])

    # the same million units as packed evidence (svt_pack_evidence -> svt_genotype_packed) agree on everything
    with hip.PackedEvidence(batch) as packed:
        assert np.array_equal(hip.genotype_packed(packed, device=hip_device).rec, first.rec)

    # the oracle on a bounded random sample of the same units
    pick = np.sort(rng.choice(n, 20_000, replace=False))
    sample = synth.permute_units(batch, pick)
    want = c_oracle.genotype_batch(sample, flags=0)
    assert_parity(ev.Results(first.rec[pick].copy()), want)

    # counts are truncations of the tallies they summarise (classic.py:455-465): int(a)+int(b) <= int(a+b)
    c = first.counts
    col = {name: c[:, i].astype(np.int64) for i, name in enumerate(ev.COUNT_NAMES)}
    called = first.gt != ev.GT_BLANK
    slack = (col["DP"] - col["RO"] - col["AO"])[called]
    assert slack.min() >= 0 and slack.max() <= 1
    t = first.tallies[called]
    assert np.array_equal(col["DP"][called], np.trunc(t[:, 0] + t[:, 1] + t[:, 2] + t[:, 3] + t[:, 4]).astype(np.int64))


# ------------------------------------------------------------------------------------------
# histogram windows: ospan_len against [key_min, key_min + n_bins) and the same shifted by var_length
# (the clamped table indices of the streaming kernel, the table codes of packed pair entries)
# ------------------------------------------------------------------------------------------
def _mode_of(batch, flags=0, device=0):
    from svtyper_amd import hip
    with hip.DeviceBatch(batch, device, flags) as d:
        return d.table_mode()


def _sweep_batch(lib, var_lengths, svtype, extra_libs=()):
    """One unit per var_length whose records sweep ospan_len across both histogram windows
    (key_min .. key_min + n_bins and the same shifted by var_length), every straddle-bit combination."""
    kmin, nb = int(lib.key_min), int(len(lib.hist))
    units, recs, off = [], [], [0]
    for vl in var_lengths:
        spans = set()
        for base in (kmin, kmin + vl):
            spans.update(range(base - 3, base + 4))
            spans.update(range(base + nb - 4, base + nb + 4))
            spans.update(range(base + nb // 2 - 8, base + nb // 2 + 8))
        spans = sorted(s for s in spans if 0 <= s < 2**31)
        n = len(spans) * 7
        r = np.zeros(n, ev.RECORD_DTYPE)
        r["ospan_len"] = np.repeat(spans, 7)
        r["flags"] = np.tile(np.arange(1, 8), len(spans)).astype(np.uint32) | np.uint32(ev.REC_HAS_PAIR)
        r["mapq_a"], r["mapq_b"] = 60, 37
        r["rs_a"] = 20
        u = np.zeros(1, ev.UNIT_DTYPE)
        u["svtype"] = svtype
        u["var_length"] = vl if svtype == 0 else 0
        u["pos_delta"] = max(vl, 10_000)          # well past the small-DEL gate
        units.append(u)
        recs.append(r)
        off.append(off[-1] + n)
    return ev.EvidenceBatch(np.asarray(off, np.uint64), np.concatenate(units), np.concatenate(recs),
                            [lib, *extra_libs], 1.0, 1.0)


@pytest.mark.parametrize("svtype", [0, 1, 2])
def test_histogram_windows(hip_device, fixture_library, svtype):
    nb = len(fixture_library.hist)
    vls = [0, 1, 37, nb - 1, nb, nb + 1, 3 * nb, 100_000, 2**29]
    batch = _sweep_batch(fixture_library, vls, svtype)
    assert _mode_of(batch) == 0
    for flags in ALL_FLAGS:
        got, want = run_both(batch, flags)
        assert_parity(got, want)


def test_histogram_windows_multi_library(hip_device, fixture_library):
    other = synth.normal_library(420.0, 95.0, seed=3)
    nb = len(other.hist)
    batch = _sweep_batch(other, [0, 5, nb - 1, nb, nb + 7, 50_000], 0, extra_libs=(fixture_library,))
    batch.records["flags"][1::2] |= np.uint32(1 << ev.REC_LIB_SHIFT)   # alternate the two libraries
    assert _mode_of(batch) == 1                     # no svt_unit.libs hints, but both libraries fit LDS: one window = the batch
    hinted = synth.permute_units(batch, np.arange(batch.n_units))
    hinted.units["libs"] = ev.unit_libs(0, 2)
    assert _mode_of(hinted) == 1                    # both libraries in every unit's window
    for flags in ALL_FLAGS:
        got, want = run_both(batch, flags)
        assert_parity(got, want)
        again, _ = run_both(hinted, flags)
        assert again.rec.tobytes() == got.rec.tobytes()


def test_small_deletion_gate_per_library(hip_device, fixture_library):
    """pos_delta < 2 sd of the *entry's* library switches its pair evidence off (classic.py:339,383)."""
    tight = synth.normal_library(300.0, 20.0, seed=11)      # 2 sd = 40
    wide = synth.normal_library(500.0, 120.0, seed=12)      # 2 sd = 240
    batch = synth.make_units(3000, 21, [tight, wide], svtype_mix=(1.0, 0, 0, 0))
    batch.units["pos_delta"] = np.resize([10, 39, 40, 41, 100, 239, 240, 241, 1000], batch.n_units)
    batch.units["var_length"] = batch.units["pos_delta"]
    hinted = synth.permute_units(batch, np.arange(batch.n_units))
    hinted.units["libs"] = ev.unit_libs(0, 2)
    assert _mode_of(batch) == 1 and _mode_of(hinted) == 1
    for flags in ALL_FLAGS:
        got, want = run_both(batch, flags)
        assert_parity(got, want)
        assert run_both(hinted, flags)[0].rec.tobytes() == got.rec.tobytes()


def test_shapes_outside_the_fast_modes_stay_exact(hip_device, fixture_library):
    """Batches the packed format / the one-library mode cannot express take another mode, silently and exactly."""
    # (a) a histogram wider than a packed pair entry's 12-bit code allows (one library: still tables in LDS)
    broad = synth.normal_library(3000.0, 900.0, seed=5)
    assert len(broad.hist) > 4095
    a = synth.make_units(1500, 31, [broad], svtype_mix=(0.6, 0.2, 0.2, 0.0))
    # (b) several libraries and a MAPQ above 127 on a pair entry
    libs = [fixture_library, synth.normal_library(420.0, 95.0, seed=3)]
    b = synth.make_units(1500, 32, libs, svtype_mix=(0.6, 0.2, 0.2, 0.0))
    b.records["mapq_a"][::11] = 200
    # (c) a unit whose records reference libraries far apart
    many = [synth.normal_library(300.0 + 10 * i, 40.0 + i, seed=40 + i) for i in range(7)]
    c = synth.make_units(1500, 33, many, svtype_mix=(0.6, 0.2, 0.2, 0.0))
    fl = c.records["flags"] & ~np.uint32(0xff << ev.REC_LIB_SHIFT)
    c.records["flags"] = fl | (np.resize([0, 6, 3], c.n_records).astype(np.uint32) << ev.REC_LIB_SHIFT)
    # (d) a negative DEL length
    d = synth.make_units(1500, 34, [fixture_library], svtype_mix=(1.0, 0, 0, 0))
    d.units["var_length"][::9] = -250
    from svtyper_amd import hip
    assert hip.PackedEvidence.try_pack(a) is None and hip.PackedEvidence.try_pack(d) is None
    assert [_mode_of(x) for x in (a, b, c, d)] == [0, 1, 1, 0]     # (b, c: no hints, the batch's libraries as one window)
    for batch in (a, b, c, d):
        for flags in ALL_FLAGS:
            got, want = run_both(batch, flags)
            assert_parity(got, want)


def test_general_tables_flag_gives_the_same_bytes(hip_device, fixture_library):
    """SVT_FLAG_GENERAL_TABLES sends any batch through the general mode (tables through L2, exact 64-bit keys), which is
    otherwise reserved for geometries the 32-bit keys cannot express: one library, both associations, edge cases."""
    from svtyper_amd import hip
    for batch in (synth.make_units(9000, 41, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), min_frags=0),
                  synth.make_edge_cases([fixture_library], seed=9)):
        for flags in ALL_FLAGS:
            with hip.DeviceBatch(batch, hip_device, flags) as d:
                assert d.table_mode() == 0
                d.genotype(sync=True)
                want = d.results().rec.tobytes()
            with hip.DeviceBatch(batch, hip_device, flags | ev.FLAG_GENERAL_TABLES) as d:
                assert d.table_mode() == 2
                d.genotype(sync=True)
                assert d.results().rec.tobytes() == want


def test_pooled_buffers_do_not_leak_state(hip_device, fixture_library):
    """svt_batch_destroy hands the big device buffers to a pool and the next create reuses them
    (larger than needed, full of the previous batch's bytes): results must not depend on that."""
    from svtyper_amd import hip
    big = synth.make_units(30_000, 71, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1))
    small = synth.make_units(17_000, 72, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=40, sd_frags=30,
                             min_frags=0, max_frags=120)
    want_small = run_both(small)[1]
    for flags in (0, ev.FLAG_SSO_ASSOCIATION, 0):
        got_big, want_big = run_both(big, flags)
        assert_parity(got_big, want_big)
        assert_parity(hip.genotype_batch(small, device=hip_device, flags=flags), want_small if not flags else run_both(small, flags)[1])
    hip.trim()
    assert_parity(hip.genotype_batch(small, device=hip_device), want_small)


def _segmented(batch, cuts):
    """the batch with its record array cut at `cuts` (record indices, any order; duplicates give empty segments)"""
    edges = [0] + sorted(int(c) for c in cuts) + [batch.n_records]
    segs = [batch.records[a:b].copy() for a, b in zip(edges[:-1], edges[1:])]
    return ev.SegmentedBatch(batch.rec_offset, batch.units, segs, batch.libs, batch.split_weight, batch.disc_weight)


def test_records_handed_over_in_segments(hip_device, fixture_library):
    """svt_batch_create_segments (ABI 16): the batch whose record array is the concatenation of the segments -- the same bytes
    as svt_batch_create over the joined array, for one library, for library windows (hinted, sample-major, site-major result
    records) and for several libraries without hints (the windows are read off the records: segments joined in scratch);
    segments cut inside units, empty segments, no segment at all for an empty batch; mismatching lengths are refused."""
    from svtyper_amd import hip
    rng = np.random.default_rng(44)
    one = synth.make_units(30000, 91, [fixture_library], mean_frags=40, sd_frags=25, min_frags=0, max_frags=200, frac_empty=0.02, frac_skip=0.01)
    multi = synth.make_multisample(700, 8, seed=17, mean_frags=30, sd_frags=12, min_frags=0, max_frags=90)
    by_sample, _ = synth.to_sample_major(multi, 8)
    unhinted = ev.EvidenceBatch(multi.rec_offset, multi.units.copy(), multi.records, multi.libs, multi.split_weight, multi.disc_weight)
    unhinted.units["libs"] = 0
    for batch, order in ((one, 0), (by_sample, 8), (multi, 0), (unhinted, 0)):
        for flags in (0, ev.FLAG_SSO_ASSOCIATION | ev.FLAG_RESULT96):
            with hip.DeviceBatch(batch, hip_device, flags) as d:
                if order:
                    d.result_order(order)
                d.genotype(sync=True)
                want, mode = d.results().rec.tobytes(), d.table_mode()
            cut_sets = [[], [batch.n_records // 2], list(rng.integers(0, batch.n_records + 1, 7)), [0, 0, batch.n_records, 5, 5]]
            if order:      # one segment per sample, as NativeUnitCollector hands them over
                n_sites = batch.n_units // order
                cut_sets.append([int(batch.rec_offset[k * n_sites]) for k in range(1, order)])
            for cuts in cut_sets:
                with hip.DeviceBatch.from_segments(_segmented(batch, cuts), hip_device, flags) as d:
                    if order:
                        d.result_order(order)
                    d.genotype(sync=True)
                    assert d.table_mode() == mode
                    assert d.results().rec.tobytes() == want, (order, flags, cuts)
    empty = one.slice(0, 0)
    with hip.DeviceBatch.from_segments(ev.SegmentedBatch(empty.rec_offset, empty.units, [], empty.libs), hip_device, 0) as d:
        d.genotype(sync=True)
        assert d.results().n_units == 0
    L = hip.load()
    import ctypes as C
    small = one.slice(0, 100)
    cb = small.as_c()
    segs = np.zeros(2, np.dtype([("records", "<u8"), ("n_records", "<u8")]))
    for lengths in ((small.n_records - 1, 0), (small.n_records, 1), (small.n_records + 5, 0)):
        segs[0] = (small.records.ctypes.data, lengths[0])
        segs[1] = (small.records.ctypes.data, lengths[1])
        h = C.c_void_p()
        assert L.svt_batch_create_segments(C.byref(cb), segs.ctypes.data, 2, hip_device, 0, C.byref(h)) == -1 and not h.value      # SVT_ERR_INVALID
        assert b"segments hold" in L.svt_last_error()


@pytest.mark.parametrize("n_samples", [1, 3, 32])
def test_site_qual_on_device(hip_device, fixture_library, n_samples):
    """svt_batch_site_qual == the reference's running QUAL (classic.py:216-217,485,498), bit for bit"""
    from svtyper_amd import hip
    batch = synth.make_edge_cases([fixture_library], seed=5)
    n_sites = batch.n_units // n_samples
    batch = batch.slice(0, n_sites * n_samples)
    rng = np.random.default_rng(n_samples)
    initial = np.where(rng.random(n_sites) < 0.5, 0.0, rng.random(n_sites) * 1000.0)
    with hip.DeviceBatch(batch, hip_device) as d:
        d.genotype(sync=True)
        res = d.results()
        got0 = d.site_qual(n_samples)
        got1 = d.site_qual(n_samples, initial)
    codes = set(np.unique(res.gt).tolist())
    assert {ev.GT_BLANK, ev.GT_SKIPPED, ev.GT_MISSING} <= codes and codes & {0, 1, 2}
    for init, got in ((np.zeros(n_sites), got0), (initial, got1)):
        want = np.zeros(n_sites)
        for s in range(n_sites):
            q = float(init[s])
            for k in range(n_samples):
                r = res.rec[s * n_samples + k]
                if r["gt"] >= 0:
                    q += float(r["sq"])
                elif r["gt"] == ev.GT_BLANK:
                    q = 0.0
            want[s] = q
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    with pytest.raises(hip.SvtyperHipError):
        with hip.DeviceBatch(batch, hip_device) as d:
            d.site_qual(n_samples)          # no pass has run yet


# ------------------------------------------------------------------------------------------
# MAPQ patterns: every mix of common / other MAPQ pairs (the one-half-word and wide pair entries of packed
# evidence, which run_both sends through svt_genotype_packed as well), MAPQ 0 / 255 on either read
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pattern", ["all_common", "all_wide", "alternate", "runs", "random", "mapq255", "vote_other"])
def test_mapq_patterns(hip_device, fixture_library, pattern):
    import zlib
    rng = np.random.default_rng(zlib.crc32(pattern.encode()))
    batch = synth.make_units(3000, 61, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=40, sd_frags=25,
                             min_frags=0)
    n = batch.n_records
    a, b = batch.records["mapq_a"], batch.records["mapq_b"]
    if pattern == "all_common":
        a[:], b[:] = 60, 60
    elif pattern == "all_wide":
        a[:] = rng.integers(1, 60, n); b[:] = 60 - a // 2           # never (60, 60), all different
    elif pattern == "alternate":
        a[:], b[:] = 60, 60
        a[::2] = 37
    elif pattern == "runs":
        a[:], b[:] = 60, 60
        k = np.arange(n)
        wide = (k % 11) < 3                                          # 3 wide, 8 common, ...: every alignment case
        a[wide], b[wide] = 23, 59
    elif pattern == "random":
        a[:] = rng.choice([60, 60, 60, 0, 1, 40, 255], n); b[:] = rng.choice([60, 60, 60, 0, 13, 255], n)
    elif pattern == "mapq255":
        a[:], b[:] = 255, 255
        a[::5] = 128
    else:   # the vote picks (40, 13); (60, 60) entries are then the wide ones
        a[:], b[:] = 40, 13
        a[::4], b[::4] = 60, 60
    for flags in ALL_FLAGS:
        got, want = run_both(batch, flags)
        assert_parity(got, want)


def test_sum_of_likelihoods_across_the_underflow_band(hip_device, fixture_library):
    """gt_sum = sum(10**GL) (classic.py:473-481) from comfortable magnitudes down through the subnormal range to
    0: the kernel forms 10**x with exp10 while the largest term is far from underflow and with pow below that, SQ
    must stay within 1e-6 of the reference arithmetic on both sides of the switch and GT './.' must start at the
    same unit."""
    # alt-only pair evidence: QA = n (or n - 1), QR = 0; GL of the best genotype = QA * log10(0.9) for a DEL
    # (-0.046 per read) and QA * log10(1/3) for a DUP (-0.477 per read): both sweep -250 .. -335
    for svtype, counts in ((0, np.arange(5500, 7320, 3)), (1, np.arange(520, 710))):
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        r = np.zeros(int(off[-1]), ev.RECORD_DTYPE)
        r["flags"] = np.uint32(ev.REC_ALT_STRADDLE | ev.REC_HAS_PAIR)
        r["mapq_a"], r["mapq_b"] = 60, 60
        r["ospan_len"] = 100_000                      # far outside the histogram
        u = np.zeros(len(counts), ev.UNIT_DTYPE)
        u["svtype"] = svtype
        u["var_length"] = 5000 if svtype == 0 else 0
        u["pos_delta"] = 5000
        batch = ev.EvidenceBatch(off, u, r, [fixture_library], 1.0, 1.0)
        for flags in ALL_FLAGS:
            got, want = run_both(batch, flags)
            assert_parity(got, want)
        best = want.gl.max(axis=1)
        called = want.gt >= 0
        assert (best[called] > -280).any() and ((best[called] < -295) & (best[called] > -310)).any()
        assert (want.gt == ev.GT_MISSING).any() and called.any()
        # the band where 10**GL is subnormal is covered unit by unit
        assert ((best < -308) & (best > -324)).sum() >= 5


# ------------------------------------------------------------------------------------------
# BASELINE.json configs[4] at its own per-GPU size: 500 k sites x 32 samples over 8 GPUs = 62 500 sites x 32 samples
# = 2 M units of ~100 fragment records (200 M records, 3.2 GB), ~66 libraries (per-sample library windows), through
# size-independent properties (the shape classic.py:279 iterates with the per-sample libraries of parsers.py:432-447)
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_c5():
    import multiprocessing as mp
    with mp.get_context("fork").Pool(16) as pool:
        return synth.make_multisample(62_500, 32, synth.BASE_SEED + 5, pool_map=pool.map)


def test_full_size_multisample_properties(hip_device, full_c5):
    from oracle import c_oracle
    from svtyper_amd import hip
    batch = full_c5
    n, S = batch.n_units, 32
    assert n == 62_500 * S == 2_000_000 and len(batch.libs) >= 32 and batch.n_records > 190_000_000
    assert (batch.units["sample"][: 2 * S] == np.tile(np.arange(S), 2)).all()      # site-major
    with hip.DeviceBatch(batch, device=hip_device) as d:
        assert d.layout_name() == "stream"
        d.genotype(sync=True)
        first = d.results()
        qual_dev = d.site_qual(S)
        d.genotype(sync=True)                                    # idempotence
        assert np.array_equal(d.results().rec, first.rec)
    base = _digest(first)
    assert not first.rec["pad"].any()
    # QUAL over a site's samples: device kernel == the running host sum (classic.py:485,498), bit for bit
    assert np.array_equal(qual_dev.view(np.uint64), hip.site_qual_host(first, S).view(np.uint64))
    # split invariance at a site boundary, and the multi-device entry with group = samples per site
    cut = 20_833 * S
    lo = hip.genotype_batch(batch.slice(0, cut), device=hip_device)
    hi = hip.genotype_batch(batch.slice(cut, n), device=hip_device)
    assert np.array_equal(np.concatenate([_digest(lo), _digest(hi)]), base)
    multi = hip.genotype_multi(batch, [hip_device] * 3, group=S)
    assert np.array_equal(_digest(multi), base)
    assert all(lo_ % S == 0 for lo_, _ in hip.shard_bounds(batch.rec_offset, 3, S))
    # without the svt_unit.libs hints svt_batch_create reads the windows off the uploaded records (svt_window_scan_kernel);
    # the pipelined one-shot and SVT_FLAG_GENERAL_TABLES take the general mode (tables through L2): the same bytes each way
    units = batch.units.copy()
    units["libs"] = 0
    plain = ev.EvidenceBatch(batch.rec_offset, units, batch.records, batch.libs, batch.split_weight, batch.disc_weight)
    with hip.DeviceBatch(plain, device=hip_device) as d:
        assert d.table_mode() == 1
        d.genotype(sync=True)
        assert np.array_equal(d.results().rec, first.rec)
    assert np.array_equal(hip.genotype_batch(plain, device=hip_device).rec, first.rec)
    with hip.DeviceBatch(batch, device=hip_device, flags=ev.FLAG_GENERAL_TABLES) as d:
        assert d.table_mode() == 2
        d.genotype(sync=True)
        assert np.array_equal(d.results().rec, first.rec)
    # the oracle on a bounded random sample of whole sites
    rng = np.random.default_rng(7)
    sites = np.sort(rng.choice(62_500, 700, replace=False))
    pick = (sites[:, None] * S + np.arange(S)[None, :]).reshape(-1)
    want = c_oracle.genotype_batch(synth.permute_units(batch, pick), flags=0)
    assert_parity(ev.Results(first.rec[pick].copy()), want)
    assert {0, 1, 2} <= set(np.unique(first.gt).tolist())


# ------------------------------------------------------------------------------------------
# streaming layout with several libraries: per-sample library windows (svt_unit.libs)
# ------------------------------------------------------------------------------------------
def test_library_windows_in_the_streaming_kernel(hip_device, fixture_library):
    """Units that say which libraries their sample owns are grouped by that window and a workgroup stages only the
    window's histograms (table mode 1); without the hint svt_batch_create derives the windows from the records on the
    device (mode 1 again, windows as narrow as the records allow); SVT_FLAG_GENERAL_TABLES keeps every table in L2 (the
    general mode, 2) -- the same bits each way."""
    from svtyper_amd import hip
    batch = synth.make_multisample(150, 32, seed=5, mean_frags=40, sd_frags=15, min_frags=5, max_frags=90)
    assert (batch.units["libs"] != 0).all() and len(np.unique(batch.units["libs"])) == 32
    for flags in (0, ev.FLAG_SSO_ASSOCIATION):
        with hip.DeviceBatch(batch, hip_device, flags) as d:
            assert d.layout_name() == "stream" and d.table_mode() == 1
            d.genotype(sync=True)
            win = d.results()
        plain = synth.permute_units(batch, np.arange(batch.n_units))
        plain.units["libs"] = 0
        with hip.DeviceBatch(plain, hip_device, flags) as d:
            assert d.table_mode() == 1
            d.genotype(sync=True)
            derived = d.results()
        assert win.rec.tobytes() == derived.rec.tobytes()
        # (the multi-device entry creates one resident batch per shard on its own host thread: each derives its own windows)
        assert hip.genotype_multi(plain, [hip_device, hip_device], group=32, flags=flags).rec.tobytes() == win.rec.tobytes()
        for b_ in (batch, plain):
            with hip.DeviceBatch(b_, hip_device, flags | ev.FLAG_GENERAL_TABLES) as d:
                assert d.table_mode() == 2
                d.genotype(sync=True)
                gen = d.results()
            assert win.rec.tobytes() == gen.rec.tobytes()
        from oracle import c_oracle
        assert_parity(win, c_oracle.genotype_batch(batch, flags=flags))
    # groups that do not fill a workgroup, one unit per window, units in any order
    order = np.random.default_rng(3).permutation(batch.n_units)[:1777]
    sub = synth.permute_units(batch, order)
    got, want = run_both(sub)
    assert_parity(got, want)
    # a window wider than what the records use is fine; a record outside its unit's window is a contract violation
    wide = synth.permute_units(batch, np.arange(640))
    wide.units["libs"] = ev.unit_libs(0, len(batch.libs))
    got, want = run_both(wide)
    assert_parity(got, want)
    bad = synth.permute_units(batch, np.arange(640))
    lo = int(ev.unit_libs_first(int(bad.units["libs"][0])))
    bad.units["libs"][0] = ev.unit_libs(lo + 1, 1) if lo + 1 < len(batch.libs) else ev.unit_libs(lo - 1, 1)
    if bad.rec_offset[1] > bad.rec_offset[0]:
        with pytest.raises(hip.SvtyperHipError) as e:
            hip.genotype_batch(bad)
        assert "lib index" in str(e.value)
    beyond = synth.permute_units(batch, np.arange(64))
    beyond.units["libs"][3] = ev.unit_libs(len(batch.libs) - 1, 5)
    with pytest.raises(hip.SvtyperHipError):
        hip.genotype_batch(beyond)
    # without hints: a record that names a library the batch does not have reaches the window scan first; it must still
    # come out as the contract violation it is, and units without records must not disturb the scan
    rogue = synth.permute_units(batch, np.arange(640))
    rogue.units["libs"] = 0
    rogue.records["flags"][int(rogue.rec_offset[7])] |= np.uint32(0xff << ev.REC_LIB_SHIFT)
    with pytest.raises(hip.SvtyperHipError) as e:
        with hip.DeviceBatch(rogue, hip_device) as d:
            d.genotype(sync=True)
    assert "lib index" in str(e.value)
    keep = np.ones(batch.n_units, bool)
    keep[::7] = False
    sparse = synth.permute_units(batch, np.arange(batch.n_units))
    sparse.units["libs"] = 0
    counts = np.diff(sparse.rec_offset.astype(np.int64)) * keep               # every seventh unit loses its records
    take = np.repeat(keep, np.diff(sparse.rec_offset.astype(np.int64)))
    sparse = ev.EvidenceBatch(np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64), sparse.units, sparse.records[take],
                              sparse.libs, sparse.split_weight, sparse.disc_weight)
    with hip.DeviceBatch(sparse, hip_device) as d:
        assert d.table_mode() == 1
        d.genotype(sync=True)
        from oracle import c_oracle
        assert_parity(d.results(), c_oracle.genotype_batch(sparse, flags=0))


# ------------------------------------------------------------------------------------------
# svt_genotype / svt_genotype_packed: upload || pass || download by unit ranges (f4)
# ------------------------------------------------------------------------------------------
def test_pipelined_one_shot_equals_the_resident_batch(hip_device, fixture_library):
    """Above 32 768 units the one-shot entry points upload the payload in pieces of whole units, launch one pass per
    piece on a second stream and -- into a page-locked output array -- download its records on a third.  Same bytes
    as create + pass + results, for pageable and page-locked outputs, both associations, records and packed slots;
    a malformed record in the LAST piece is still reported."""
    from svtyper_amd import hip
    batch = synth.make_units(120_000, 91, [fixture_library], svtype_mix=(0.6, 0.2, 0.1, 0.1), mean_frags=60, sd_frags=30,
                             min_frags=0)
    assert batch.n_records * 16 > 3 * (32 << 20)          # several 32 MB pieces
    for flags in (0, ev.FLAG_SSO_ASSOCIATION):
        with hip.DeviceBatch(batch, hip_device, flags) as d:
            d.genotype(sync=True)
            want = d.results().rec.tobytes()
        assert hip.genotype_batch(batch, hip_device, flags).rec.tobytes() == want
        pinned = hip.pinned_results(batch.n_units)
        assert hip.genotype_batch(batch, hip_device, flags, out=pinned).rec.tobytes() == want
        with hip.PackedEvidence(batch) as p:
            assert hip.genotype_packed(p, hip_device, flags).rec.tobytes() == want
            assert hip.genotype_packed(p, hip_device, flags, out=pinned).rec.tobytes() == want
    # many short units: a 64 MB piece then holds more than 221 184 units and its launch takes two tiles per wave,
    # starting at a unit range that does not begin at 0
    tiny = synth.make_units(700_000, 93, [fixture_library], svtype_mix=(0.6, 0.2, 0.1, 0.1), mean_frags=8, sd_frags=4, min_frags=0,
                            max_frags=40)
    assert tiny.n_records * 16 > (64 << 20) and tiny.n_records * 16 / 700_000 * 221_184 < (64 << 20)
    with hip.DeviceBatch(tiny, hip_device, 0) as d:
        d.genotype(sync=True)
        want = d.results().rec.tobytes()
    assert hip.genotype_batch(tiny, hip_device, 0).rec.tobytes() == want
    assert hip.genotype_batch(tiny, hip_device, 0, out=hip.pinned_results(tiny.n_units)).rec.tobytes() == want
    bad = synth.make_units(120_000, 91, [fixture_library], svtype_mix=(0.6, 0.2, 0.1, 0.1), mean_frags=60, sd_frags=30, min_frags=0)
    bad.records["flags"][bad.n_records - 3] |= 1 << 28
    with pytest.raises(hip.SvtyperHipError) as e:
        hip.genotype_batch(bad, hip_device)
    assert "reserved/undefined bits" in str(e.value)
    # several libraries with windows: one piece, one launch over the window chunks -- same entry point
    ms = synth.make_multisample(1100, 32, seed=11, mean_frags=30, sd_frags=10, min_frags=4, max_frags=60)
    assert ms.n_units >= 32768
    with hip.DeviceBatch(ms, hip_device) as d:
        d.genotype(sync=True)
        assert hip.genotype_batch(ms, hip_device).rec.tobytes() == d.results().rec.tobytes()


def test_sample_major_units_come_back_site_major(hip_device, fixture_library):
    """svt_batch_result_order: the same (site, sample) units handed over sample-major (what a producer that reads BAM by
    BAM emits; a sample's units are then contiguous in HBM) with the results scattered back site-major must give the
    bytes of the site-major batch -- in the library-window mode, without hints (derived windows; general mode), with one library,
    for both associations, for launches of one and of two tiles per wave -- and QUAL over a site's samples
    (svt_batch_site_qual reads the site-major result array) must not change either."""
    from svtyper_amd import hip
    n_samples = 32
    for n_sites, tag in ((150, "one tile"), (7500, "two tiles")):          # 240 000 units >= the two-tile threshold
        site = synth.make_multisample(n_sites, n_samples, seed=11, mean_frags=12 if n_sites > 1000 else 40, sd_frags=5,
                                      min_frags=2, max_frags=40 if n_sites > 1000 else 90)
        samp, order = synth.to_sample_major(site, n_samples)
        assert samp.n_units == site.n_units and np.array_equal(samp.units, site.units[order])
        for flags in (0, ev.FLAG_SSO_ASSOCIATION):
            with hip.DeviceBatch(site, hip_device, flags) as d:
                assert d.table_mode() == 1
                d.genotype(sync=True)
                want = d.results()
                want_q = d.site_qual(n_samples)
            with hip.DeviceBatch(samp, hip_device, flags) as d:
                assert d.table_mode() == 1
                d.genotype(sync=True)
                plain = d.results()                               # unit order = sample-major
                d.result_order(n_samples)
                d.genotype(sync=True)
                got = d.results()
                got_q = d.site_qual(n_samples)
                d.result_order(0)
                d.genotype(sync=True)
                back = d.results()
            assert plain.rec.tobytes() == want.rec[order].tobytes(), tag
            assert got.rec.tobytes() == want.rec.tobytes(), tag
            assert back.rec.tobytes() == plain.rec.tobytes(), tag
            assert got_q.tobytes() == want_q.tobytes(), tag
        if n_sites > 1000:
            continue
        # the general mode (no hints) and a one-library batch take the same scatter
        nh_site = synth.permute_units(site, np.arange(site.n_units))
        nh_site.units["libs"] = 0
        nh_samp = synth.permute_units(samp, np.arange(samp.n_units))
        nh_samp.units["libs"] = 0
        for fl, mode in ((0, 1), (ev.FLAG_GENERAL_TABLES, 2)):    # windows derived from the records / the general mode
            with hip.DeviceBatch(nh_samp, hip_device, fl) as d:
                assert d.table_mode() == mode
                d.result_order(n_samples)
                d.genotype(sync=True)
                assert d.results().rec.tobytes() == want0(nh_site, hip_device).rec.tobytes()
    one = synth.make_units(640, 3, [fixture_library])
    o_samp, o_order = synth.to_sample_major(one, 4)
    with hip.DeviceBatch(o_samp, hip_device, 0) as d:
        assert d.table_mode() == 0
        d.result_order(4)
        d.genotype(sync=True)
        assert d.results().rec.tobytes() == want0(one, hip_device).rec.tobytes()
        with pytest.raises(hip.SvtyperHipError):
            d.result_order(7)                                     # 640 units are not a multiple of 7


def want0(batch, device):
    from svtyper_amd import hip
    with hip.DeviceBatch(batch, device, 0) as d:
        d.genotype(sync=True)
        return d.results()
