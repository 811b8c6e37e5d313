"""SVT_FLAG_RESULT96 (ABI 13): the pass writes the 96-byte result record of SURVEY.md 8(d) -- GL, SQ, the five tallies,
QR / QA / GQ, GT -- tagged with its unit and in the order the kernel finishes the units (whole cache lines per wave); the
host puts every record where its tag says and restores DP, RO, AO, RS, AS, ASC, RP, AP from the tallies as the reference
computes them (svtyper/classic.py:455-469: int() of the sums).  The host-side results must be the same bytes as the
128-byte path's."""
import numpy as np
import pytest

from svtyper_amd import evidence as ev, synth


def _to96(rec128, units=None):
    r = np.zeros(len(rec128), ev.RESULT96_DTYPE)
    for f in ("gl", "sq", "tallies", "gt"):
        r[f] = rec128[f]
    r["qr"], r["qa"], r["gq"] = rec128["counts"][:, 0], rec128["counts"][:, 1], rec128["counts"][:, 2]
    r["unit"] = np.arange(len(rec128)) if units is None else units
    return r


@pytest.mark.parametrize("sso", [0, ev.FLAG_SSO_ASSOCIATION])
def test_expansion_restores_the_oracles_counts(fixture_library, sso):
    """host only: the oracle's 128-byte records, cut down to the 96-byte form, come back bit for bit from
    svt_results_expand96 -- blank, skipped, './.' and called units alike (the derivation rule against the reference's
    restatement on every kind of unit)."""
    from oracle import c_oracle
    from svtyper_amd import hip
    parts = [synth.make_edge_cases([fixture_library], seed=5),
             synth.make_units(6000, 17, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=40, sd_frags=30, min_frags=0,
                              frac_empty=0.05, frac_skip=0.03)]
    for batch in parts:
        want = c_oracle.genotype_batch(batch, flags=sso)
        assert {ev.GT_BLANK, ev.GT_SKIPPED} <= set(np.unique(want.gt).tolist())
        n = want.n_units
        got = hip.expand96(_to96(want.rec), n)
        assert got.rec.tobytes() == want.rec.tobytes()
        # the records in any order, padding records between them: every one lands where its tag says
        rng = np.random.default_rng(3)
        order = rng.permutation(n)
        shuffled = _to96(want.rec[order], units=order)
        padded = np.zeros(n + 300, ev.RESULT96_DTYPE)
        padded["unit"] = ev.NO_UNIT
        at = np.sort(rng.choice(n + 300, n, replace=False))
        padded[at] = shuffled
        assert hip.expand96(padded, n).rec.tobytes() == want.rec.tobytes()
        into = np.zeros(n, ev.RESULT_DTYPE)
        hip.expand96(padded, n, into)
        assert into.tobytes() == want.rec.tobytes()
        # a unit missing, a unit twice, a tag beyond the units: refused
        # ("pair": one unit twice AND another one missing at once -- {1, 1, 2, 2} for {0, 1, 2, 3} keeps both the count and the
        # sum of the tags, which was all the first form of the check looked at)
        for breaker in ("missing", "twice", "beyond", "pair"):
            x = padded.copy()
            if breaker == "missing":
                x["unit"][at[5]] = ev.NO_UNIT
            elif breaker == "twice":
                x["unit"][at[5]] = x["unit"][at[6]]
            elif breaker == "pair":
                lo, hi = np.nonzero(x["unit"] == 0)[0][0], np.nonzero(x["unit"] == 3)[0][0]
                x["unit"][lo], x["unit"][hi] = 1, 2
            else:
                x["unit"][at[5]] = n
            with pytest.raises(hip.SvtyperHipError):
                hip.expand96(x, n)
    assert hip.expand96(np.zeros(0, ev.RESULT96_DTYPE), 0).n_units == 0
    with pytest.raises(ValueError):
        hip.expand96(np.zeros(100, np.uint8), 1)


@pytest.mark.gpu
def test_device_records_are_the_96_byte_form(hip_device, fixture_library):
    """what the kernel leaves in HBM under the flag: result_slots() tagged svt_result96 records -- every unit exactly once, whole
    workgroups (padding tagged NO_UNIT), fields equal to the 128-byte pass's; the host entry points (resident batch, pageable
    and page-locked one-shot, packed evidence, several devices) return the 128-byte records unchanged"""
    import ctypes as C
    from svtyper_amd import hip
    batch = synth.make_units(70_000, 23, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=30, sd_frags=20, min_frags=0,
                             frac_empty=0.03, frac_skip=0.02)
    for sso in (0, ev.FLAG_SSO_ASSOCIATION):
        want = hip.genotype_batch(batch, hip_device, sso)
        flags = sso | ev.FLAG_RESULT96
        with hip.DeviceBatch(batch, hip_device, flags) as d:
            assert d.result_bytes() == 96
            d.genotype(sync=True)
            assert d.results().rec.tobytes() == want.rec.tobytes()
            slots = d.result_slots()
            # (whole 64-unit tiles: 70 000 units are less than one round of workgroups and take the cooperative kernel, whose
            # workgroups hold 64 ... 256 units; the streaming kernel's hold 256 or 512)
            assert batch.n_units <= slots < batch.n_units + 512 and slots % 64 == 0
            raw = np.zeros(slots, ev.RESULT96_DTYPE)
            lib = hip.load()
            lib.svt_debug_copy_to_host.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
            hip._check(lib.svt_debug_copy_to_host(hip_device, C.c_void_p(raw.ctypes.data), C.c_void_p(d.device_results_ptr()), raw.nbytes))
            real = raw[raw["unit"] != ev.NO_UNIT]
            assert len(real) == batch.n_units and np.array_equal(np.sort(real["unit"]), np.arange(batch.n_units))
            assert not raw["pad"].any() and not raw["pad2"].any()
            assert real[np.argsort(real["unit"])].tobytes() == _to96(want.rec).tobytes()
            # the kernel's order: a wave's 64 records are units of similar length (the workgroup's length sort), not neighbours
            assert (np.diff(real["unit"][:64].astype(np.int64)) != 1).any()
        with hip.DeviceBatch(batch, hip_device, sso) as d:
            assert d.result_bytes() == 128
        # one shot (upload || pass || download by unit ranges), output in pageable and in page-locked memory
        assert hip.genotype_batch(batch, hip_device, flags).rec.tobytes() == want.rec.tobytes()
        pinned = hip.pinned_results(batch.n_units)
        assert hip.genotype_batch(batch, hip_device, flags, out=pinned).rec.tobytes() == want.rec.tobytes()
        with hip.PackedEvidence(batch) as p:
            assert hip.genotype_packed(p, hip_device, flags).rec.tobytes() == want.rec.tobytes()
            assert hip.genotype_packed(p, hip_device, flags, out=pinned).rec.tobytes() == want.rec.tobytes()
            with hip.DeviceBatch.from_packed(p, hip_device, flags) as dp:
                assert dp.result_bytes() == 96
                dp.genotype(sync=True)
                assert dp.results().rec.tobytes() == want.rec.tobytes()
        assert hip.genotype_multi(batch, [hip_device, hip_device, hip_device], flags=flags).rec.tobytes() == want.rec.tobytes()


@pytest.mark.gpu
def test_site_qual_and_result_order_on_96_byte_records(hip_device):
    """QUAL over a site's samples reads SQ / GT out of the 96-byte records; sample-major units land site-major"""
    from svtyper_amd import hip
    multi = synth.make_multisample(300, 8, seed=13, mean_frags=30, sd_frags=10, min_frags=0, max_frags=70)
    by_sample, _ = synth.to_sample_major(multi, 8)
    with hip.DeviceBatch(multi, hip_device, 0) as d:
        d.genotype(sync=True)
        want, q_want = d.results().rec.tobytes(), d.site_qual(8)
    init = np.linspace(0.0, 5.0, 300)
    with hip.DeviceBatch(multi, hip_device, ev.FLAG_RESULT96) as d:
        d.genotype(sync=True)
        assert d.results().rec.tobytes() == want and d.site_qual(8).tobytes() == q_want.tobytes()
        q_init = d.site_qual(8, initial=init)
    with hip.DeviceBatch(by_sample, hip_device, ev.FLAG_RESULT96) as d:
        d.result_order(8)
        d.genotype(sync=True)
        assert d.results().rec.tobytes() == want and d.site_qual(8).tobytes() == q_want.tobytes()
        assert d.site_qual(8, initial=init).tobytes() == q_init.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("n_samples", [1, 3, 32])
def test_site_qual_over_tagged_records_stays_on_the_device(hip_device, n_samples):
    """SVT_FLAG_RESULT96: svt_batch_site_qual scatters (SQ, GT) by tag and sums per site on the device (svt_site_qual_scatter_kernel /
    svt_site_qual_entries_kernel) -- the reference's running QUAL (classic.py:216-217,485,498) bit for bit, with padding slots,
    blank and skipped samples, site-major and sample-major input, with and without incoming QUAL; and it is a few launches,
    not a download of every record (60 k units: well under the 128-byte records' PCIe time)."""
    import time
    from svtyper_amd import hip
    n_sites = 60000 // n_samples
    multi = synth.make_multisample(n_sites, n_samples, seed=29, mean_frags=12, sd_frags=8, min_frags=0, max_frags=40)
    init = np.linspace(-1.0, 9.0, n_sites)
    with hip.DeviceBatch(multi, hip_device, ev.FLAG_RESULT96) as d:
        d.genotype(sync=True)
        res = d.results()
        assert d.result_slots() >= multi.n_units
        for initial in (None, init):
            want = np.asarray(hip.site_qual_host(res, n_samples, initial))
            got = d.site_qual(n_samples, initial)
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
        assert (res.rec["gt"] < 0).any() and (res.rec["gt"] >= 0).any()
        d.site_qual(n_samples)
        t0 = time.perf_counter()
        for _ in range(5):
            d.site_qual(n_samples)
        per_call_ms = (time.perf_counter() - t0) / 5 * 1e3
        print("site_qual over %d tagged records: %.3f ms per call" % (multi.n_units, per_call_ms))
    # the same records written by the packed-evidence kernel (tagged the same way)
    try:
        packed = hip.PackedEvidence(multi)
    except hip.SvtyperHipError:
        packed = None
    if packed is not None:
        with packed, hip.DeviceBatch.from_packed(packed, hip_device, ev.FLAG_RESULT96) as d:
            d.genotype(sync=True)
            assert np.array_equal(d.site_qual(n_samples, init).view(np.uint64), np.asarray(hip.site_qual_host(res, n_samples, init)).view(np.uint64))
    if n_samples > 1:
        by_sample, _ = synth.to_sample_major(multi, n_samples)
        with hip.DeviceBatch(by_sample, hip_device, ev.FLAG_RESULT96) as d:
            d.result_order(n_samples)
            d.genotype(sync=True)
            assert np.array_equal(d.site_qual(n_samples, init).view(np.uint64), np.asarray(hip.site_qual_host(res, n_samples, init)).view(np.uint64))


@pytest.mark.gpu
def test_workgroup_plan_does_not_change_the_results(hip_device, fixture_library):
    """A launch of more than one round of the chip's resident workgroups is cut into EQUAL workgroups that fill whole rounds
    (svtyper_hip.hip: wg_plan; StreamArgs.units_per_wg), the window mode cuts its chunks by the same rule: the bytes are those of
    full workgroups -- both record forms, both associations, library windows -- and of any other cut (odd workgroup sizes forced
    through the debug hook); the tagged records grow by the emptier workgroups' padding slots and still hold every unit once."""
    import ctypes as C
    from svtyper_amd import hip
    lib = hip.load()
    lib.svt_debug_wg_balance.argtypes = [C.c_int]
    lib.svt_debug_force_wg.argtypes = [C.c_uint32, C.c_uint32]
    one = synth.make_units(700_000, 31, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=12, sd_frags=8, min_frags=0,
                           frac_empty=0.02, frac_skip=0.01)
    c5 = synth.to_sample_major(synth.make_multisample(16_000, 32, seed=17, mean_frags=10, sd_frags=6, min_frags=0, max_frags=40), 32)[0]

    def run(batch, flags, order):
        with hip.DeviceBatch(batch, hip_device, flags) as d:
            if order:
                d.result_order(order)
            d.genotype(sync=True)
            return d.results().rec.tobytes(), d.result_slots()

    prev = lib.svt_debug_wg_balance(0)
    try:
        for batch, order in ((one, 0), (c5, 32)):
            for flags in (0, ev.FLAG_RESULT96, ev.FLAG_SSO_ASSOCIATION | ev.FLAG_RESULT96):
                lib.svt_debug_wg_balance(0)
                want, slots_full = run(batch, flags, order)
                lib.svt_debug_wg_balance(50)
                got, slots = run(batch, flags, order)
                assert got == want
                if flags & ev.FLAG_RESULT96:
                    assert slots % 256 == 0 and slots > slots_full >= batch.n_units          # (both batches need more than one round: the plan really cut them differently)
                else:
                    assert slots == batch.n_units
                if order == 0:
                    for per_wg, tiles in ((489, 2), (257, 2), (65, 1), (200, 1)):
                        lib.svt_debug_force_wg(per_wg, tiles)
                        try:
                            assert run(batch, flags, order)[0] == want, (per_wg, tiles)
                        finally:
                            lib.svt_debug_force_wg(0, 0)
    finally:
        lib.svt_debug_wg_balance(prev)


@pytest.mark.gpu
def test_tune_placement_keeps_the_results(hip_device, fixture_library):
    """svt_batch_tune_placement auditions freshly allocated result and record buffers with the real pass and keeps the fastest:
    whatever it keeps, the batch returns the same bytes afterwards (both record forms, library windows, packed evidence, a tiny
    and an empty batch), reports the pass time before and after (after <= before), refuses a batch whose result records are
    bound to a caller's buffer or are out as a torch view, and the next batch of the size inherits the buffers from the pool."""
    from svtyper_amd import hip
    one = synth.make_units(60_000, 41, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=30, sd_frags=20, min_frags=0,
                           frac_empty=0.02, frac_skip=0.01)
    c5 = synth.to_sample_major(synth.make_multisample(1500, 8, seed=19, mean_frags=20, sd_frags=10, min_frags=0, max_frags=60), 8)[0]
    for batch, order in ((one, 0), (c5, 8), (one.slice(0, 3), 0), (one.slice(0, 0), 0)):
        for flags in (0, ev.FLAG_RESULT96, ev.FLAG_SSO_ASSOCIATION | ev.FLAG_RESULT96):
            with hip.DeviceBatch(batch, hip_device, flags) as d:
                if order:
                    d.result_order(order)
                d.genotype(sync=True)
                want = d.results().rec.tobytes()
                r = d.tune_placement(5, 2)
                assert r["after_ms"] <= r["before_ms"] and (batch.n_units == 0 or r["before_ms"] > 0)
                assert d.results().rec.tobytes() == want          # (the records of the tuner's last pass)
                d.genotype(sync=True)
                assert d.results().rec.tobytes() == want
                if order:
                    assert d.site_qual(order).tobytes() == d.site_qual(order).tobytes()
            with hip.DeviceBatch(batch, hip_device, flags) as d:   # the pool hands the kept buffers to the next batch
                if order:
                    d.result_order(order)
                d.genotype(sync=True)
                assert d.results().rec.tobytes() == want
    with hip.PackedEvidence(one) as p, hip.DeviceBatch.from_packed(p, hip_device, ev.FLAG_RESULT96) as d:
        d.genotype(sync=True)
        want = d.results().rec.tobytes()
        r = d.tune_placement(4, 4)                                # (packed evidence: result candidates only)
        assert r["after_ms"] <= r["before_ms"]
        d.genotype(sync=True)
        assert d.results().rec.tobytes() == want
    with hip.DeviceBatch(one, hip_device, 0) as d:
        d.genotype(sync=True)
        other = hip.DeviceBatch(one, hip_device, 0)
        try:
            d.bind_device_results(other.device_results_ptr())
            with pytest.raises(hip.SvtyperHipError):
                d.tune_placement(2, 0)
            d.bind_device_results(0)
            with pytest.raises(hip.SvtyperHipError):
                d.tune_placement(65, 0)
            import weakref

            class View:      # (what device_results_tensor() registers; torch itself needs a process of its own, test_multi_device.py)
                pass
            view = View()
            d._views = [weakref.ref(view)]
            with pytest.raises(hip.SvtyperHipError):
                d.tune_placement(2, 0)
            del view
            d.tune_placement(2, 1)
        finally:
            other.close()


@pytest.mark.gpu
def test_bind_with_a_capacity_refuses_a_short_buffer(hip_device, fixture_library):
    """svt_batch_bind_device_results2: the buffer for the device records is result_slots() * result_bytes() bytes for both
    record forms; a shorter one is refused instead of being overrun by the pass."""
    import ctypes as C
    from svtyper_amd import hip
    lib = hip.load()
    lib.svt_debug_device_alloc.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.svt_debug_device_free.argtypes = [C.c_int, C.c_void_p]
    batch = synth.make_units(3000, 77, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=20, sd_frags=10, min_frags=0)
    want = hip.genotype_batch(batch, hip_device, 0)
    for flags in (0, ev.FLAG_RESULT96):
        with hip.DeviceBatch(batch, hip_device, flags) as d:
            need = d.result_slots() * d.result_bytes()
            assert need >= batch.n_units * d.result_bytes()
            buf = C.c_void_p()
            hip._check(lib.svt_debug_device_alloc(hip_device, need + 256, 0, C.byref(buf)))
            try:
                base = (buf.value + 127) // 128 * 128
                with pytest.raises(hip.SvtyperHipError):
                    d.bind_device_results(base, need - 1)
                d.bind_device_results(base, need)
                d.genotype(sync=True)
                assert d.results().rec.tobytes() == want.rec.tobytes()
                d.bind_device_results(0)
                d.genotype(sync=True)
                assert d.results().rec.tobytes() == want.rec.tobytes()
            finally:
                hip._check(lib.svt_debug_device_free(hip_device, buf))


def test_compact_gather_records_carry_the_genotype_fields(fixture_library):
    """host only: distributed.compact_tagged_records cuts tagged 96-byte records down to the 48 bytes a consumer of genotypes
    reads (GL, SQ, QR, QA, GQ, GT + the tag); results_from_compact puts two ranks' records -- any order, padding between them --
    back in unit order with exactly those fields of the oracle's records, and refuses a rank whose records do not cover it."""
    import torch
    from oracle import c_oracle
    from svtyper_amd import distributed as D
    batch = synth.make_edge_cases([fixture_library], seed=9)
    want = c_oracle.genotype_batch(batch, flags=0).rec
    n = len(want)
    cut = n // 3
    rng = np.random.default_rng(1)
    parts, sizes, counts = [], [], []
    for lo, hi in ((0, cut), (cut, n)):
        order = rng.permutation(hi - lo)
        tagged = _to96(want[lo:hi][order], units=order)
        padded = np.zeros(hi - lo + 70, ev.RESULT96_DTYPE)
        padded["unit"] = ev.NO_UNIT
        padded[np.sort(rng.choice(len(padded), hi - lo, replace=False))] = tagged
        c = D.compact_tagged_records(torch.from_numpy(padded.view(np.uint8).copy()))
        assert c.numel() == 48 * len(padded)
        parts.append(c)
        sizes.append(int(c.numel()))
        counts.append(hi - lo)
    got = D.results_from_compact(torch.cat(parts), sizes, counts).rec
    for f in ("gl", "sq", "gt"):
        assert np.array_equal(got[f], want[f])
    assert np.array_equal(got["counts"][:, :3], want["counts"][:, :3])
    assert not got["tallies"].any() and not got["counts"][:, 3:].any()
    broken = parts[0].clone()
    broken.view(-1, 48)[np.nonzero(parts[0].view(-1, 48)[:, 44:48].numpy().view(np.uint32).ravel() == 0)[0][0], 44:48] = 0xFF   # unit 0 -> padding
    with pytest.raises(ValueError):
        D.results_from_compact(torch.cat([broken, parts[1]]), sizes, counts)


@pytest.mark.gpu
def test_batch_sync_reports_what_the_enqueued_passes_found(hip_device, fixture_library):
    """svt_batch_sync (ABI 14): passes enqueued without waiting, then ONE call that waits and reports -- the results are there,
    and a record that breaks the contract (a straddle bit without HAS_PAIR) comes back as the error svt_batch_genotype(b, 1)
    would have raised."""
    from svtyper_amd import hip
    batch = synth.make_units(5000, 91, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=20, sd_frags=10, min_frags=0)
    want = hip.genotype_batch(batch, hip_device, 0)
    with hip.DeviceBatch(batch, hip_device, ev.FLAG_RESULT96) as d:
        d.genotype(sync=False)
        d.genotype_n(3)
        d.synchronize()
        assert d.results().rec.tobytes() == want.rec.tobytes()
    bad = synth.make_units(5000, 91, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=20, sd_frags=10, min_frags=1)
    bad.records["flags"][7] = (bad.records["flags"][7] | ev.REC_ALT_STRADDLE) & ~np.uint32(ev.REC_HAS_PAIR)
    with hip.DeviceBatch(bad, hip_device, 0) as d:
        d.genotype(sync=False)
        with pytest.raises(hip.SvtyperHipError):
            d.synchronize()
