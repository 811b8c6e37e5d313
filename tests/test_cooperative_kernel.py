"""Launches of less than one round of resident workgroups take svt_coop_kernel (svtyper_amd/csrc/svt_coop_kernel.h): eight
producer waves per workgroup do the table look-ups of the records, three consumer waves add the addends in record order.  Same
operands, same operations, same order per accumulator as the streaming kernel's one lane per unit -- so the same BYTES, which
is what these tests demand: against the streaming kernel (the debug hook switches the cooperative one off) and against the
oracle, both associations, both device record forms, every workgroup size, ragged / empty / skipped units, long units."""
import ctypes as C

import numpy as np
import pytest

from svtyper_amd import evidence as ev
from svtyper_amd import synth

pytestmark = pytest.mark.gpu

FLAGS = [0, ev.FLAG_SSO_ASSOCIATION, ev.FLAG_RESULT96, ev.FLAG_SSO_ASSOCIATION | ev.FLAG_RESULT96]
DEFAULT_MAX = 1 << 40        # (svt_debug_coop: no cap of its own; the rule is per CU, svtyper_hip.hip: wg_plan)


def _hook():
    from svtyper_amd import hip
    lib = hip.load()
    lib.svt_debug_coop.argtypes = [C.c_uint64, C.c_uint32]
    lib.svt_debug_coop.restype = None
    return lib


def _resident(batch, device, flags, order=0):
    from svtyper_amd import hip
    with hip.DeviceBatch(batch, device, flags) as d:
        if order:
            d.result_order(order)
        d.genotype(sync=True)
        return d.results().rec.tobytes(), d.result_slots()


def test_same_bytes_as_the_streaming_kernel_and_the_oracle(hip_device, fixture_library):
    from oracle import c_oracle
    from svtyper_amd import hip
    lib = _hook()
    batches = [
        synth.make_edge_cases([fixture_library], seed=23),                                   # empty / skipped / continuation records / 1 300-record units
        synth.make_units(9_000, 5, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=60, sd_frags=50, min_frags=0, max_frags=400,
                         frac_empty=0.04, frac_skip=0.02),
        synth.make_units(20_011, 6, [fixture_library], svtype_mix=(0.7, 0.15, 0.15, 0.0)),  # the configs[2] shape, a last workgroup that is not full
        synth.make_units(1, 7, [fixture_library]),
        synth.make_units(65, 8, [fixture_library], mean_frags=3, sd_frags=2, min_frags=0, max_frags=9),
    ]
    try:
        for batch in batches:
            for flags in FLAGS:
                lib.svt_debug_coop(0, 0)
                want, slots_stream = _resident(batch, hip_device, flags)
                oracle = c_oracle.genotype_batch(batch, flags=flags & ev.FLAG_SSO_ASSOCIATION)
                assert np.array_equal(np.frombuffer(want, ev.RESULT_DTYPE)["gt"], oracle.gt)
                for per_wg in (0, 64, 128, 192, 256):
                    lib.svt_debug_coop(1 << 40, per_wg)
                    got, slots = _resident(batch, hip_device, flags)
                    assert got == want, (batch.n_units, flags, per_wg)
                    if flags & ev.FLAG_RESULT96:
                        assert slots % 64 == 0 and batch.n_units <= slots <= batch.n_units + 255 + 256 * (per_wg == 0)
                    else:
                        assert slots == batch.n_units
                    # the one-shot entry (upload || pass || download by unit ranges: every range its own cooperative launch)
                    assert hip.genotype_batch(batch, hip_device, flags).rec.tobytes() == want
    finally:
        lib.svt_debug_coop(DEFAULT_MAX, 0)


def test_lanes_per_unit_kernels_same_bytes(hip_device, fixture_library):
    """svt_split_kernel (svtyper_amd/csrc/svt_split_kernel.h): a unit on 2 or 4 adjacent lanes, the sums handed from lane to lane
    in record order by DPP row shifts -- the streaming kernel's bytes, both associations, both record forms, ragged / empty /
    skipped / 1 300-record units, continuation records, a last workgroup that is not full, the one-shot entry."""
    from svtyper_amd import hip
    lib = _hook()
    lib.svt_debug_small_kind.argtypes = [C.c_int]
    batches = [
        synth.make_edge_cases([fixture_library], seed=29),
        synth.make_units(9_000, 15, [fixture_library], svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=60, sd_frags=50, min_frags=0, max_frags=400,
                         frac_empty=0.04, frac_skip=0.02),
        synth.make_units(40_011, 16, [fixture_library], svtype_mix=(0.7, 0.15, 0.15, 0.0)),
        synth.make_units(3, 17, [fixture_library]),
    ]
    try:
        for batch in batches:
            for flags in FLAGS:
                lib.svt_debug_small_kind(1)
                want, slots_stream = _resident(batch, hip_device, flags)
                for kind in (3, 4):
                    lib.svt_debug_small_kind(kind)
                    got, slots = _resident(batch, hip_device, flags)
                    assert got == want, (batch.n_units, flags, kind)
                    assert slots == slots_stream
                    assert hip.genotype_batch(batch, hip_device, flags).rec.tobytes() == want
    finally:
        lib.svt_debug_small_kind(0)


def test_lanes_per_unit_kernels_over_library_windows(hip_device, fixture_library):
    """the same kernels over per-sample library windows (svt_unit.libs; windows of 1-3 libraries, a window of 12, units in any
    order, hint-less batches whose windows are read off the records): the one-tile window kernel's bytes, sample-major units with
    site-major records included"""
    from svtyper_amd import hip
    lib = _hook()
    lib.svt_debug_small_kind.argtypes = [C.c_int]
    rng = np.random.default_rng(5)
    multi = synth.make_multisample(300, 8, seed=21, mean_frags=30, sd_frags=15, min_frags=0, max_frags=90)
    shuffled = synth.permute_units(multi, rng.permutation(multi.n_units)[: int(multi.n_units * 0.8)])
    wide = synth.make_units(6_000, 31, [fixture_library] + [synth.normal_library(300.0 + 20 * i, 40.0 + 3 * i, seed=40 + i) for i in range(11)],
                            svtype_mix=(0.5, 0.2, 0.2, 0.1), mean_frags=25, sd_frags=10, min_frags=0)
    by_sample, _ = synth.to_sample_major(multi, 8)
    nohint = ev.EvidenceBatch(multi.rec_offset, multi.units.copy(), multi.records, multi.libs, multi.split_weight, multi.disc_weight)
    nohint.units["libs"] = 0
    try:
        for batch, order in ((multi, 0), (shuffled, 0), (wide, 0), (by_sample, 8), (nohint, 0)):
            for flags in FLAGS:
                lib.svt_debug_small_kind(1)
                want, slots_stream = _resident(batch, hip_device, flags, order)
                for kind in (3, 4):
                    if kind == 3 and (flags & ev.FLAG_SSO_ASSOCIATION):
                        continue        # (two lanes per unit: classic association only)
                    lib.svt_debug_small_kind(kind)
                    got, slots = _resident(batch, hip_device, flags, order)
                    assert got == want, (batch.n_units, flags, kind, order)
                    assert slots == slots_stream
                lib.svt_debug_small_kind(0)
                got, _ = _resident(batch, hip_device, flags, order)
                assert got == want
    finally:
        lib.svt_debug_small_kind(0)


def test_the_default_rule_picks_it_for_small_launches_only(hip_device, fixture_library):
    """one library, by units per CU (svtyper_hip.hip: wg_plan): <= 64 cooperative (tagged records in whole 64-unit tiles);
    above: 4 / 2 lanes per unit or the streaming kernel, whole 256-unit workgroups all of them; several libraries: the streaming
    kernel (library windows)."""
    small = synth.make_units(10_000, 9, [fixture_library], mean_frags=8, sd_frags=4, min_frags=0, max_frags=20)
    large = synth.make_units(70_000, 10, [fixture_library], mean_frags=8, sd_frags=4, min_frags=0, max_frags=20)
    two = synth.make_units(10_000, 11, [fixture_library, synth.normal_library(420.0, 95.0, seed=3)], mean_frags=8, sd_frags=4, min_frags=0, max_frags=20)
    _, s_small = _resident(small, hip_device, ev.FLAG_RESULT96)
    _, s_large = _resident(large, hip_device, ev.FLAG_RESULT96)
    _, s_two = _resident(two, hip_device, ev.FLAG_RESULT96)
    assert s_small == (10_000 + 63) // 64 * 64          # 64 units per workgroup: 157 workgroups, one 64-unit tile each
    assert s_large == (70_000 + 255) // 256 * 256
    assert s_two % 256 == 0


def test_sample_major_units_site_major_records(hip_device, fixture_library):
    """svt_batch_result_order through the cooperative kernel: one library, units sample-major, records (and tags) site-major"""
    lib = _hook()
    n_sites, n_samples = 900, 8
    batch = synth.make_units(n_sites * n_samples, 12, [fixture_library], svtype_mix=(0.6, 0.2, 0.2, 0.0), mean_frags=25, sd_frags=10, min_frags=0)
    try:
        for flags in (0, ev.FLAG_RESULT96):
            lib.svt_debug_coop(0, 0)
            want, _ = _resident(batch, hip_device, flags, order=n_samples)
            lib.svt_debug_coop(1 << 40, 0)
            got, _ = _resident(batch, hip_device, flags, order=n_samples)
            assert got == want
            plain, _ = _resident(batch, hip_device, flags)
            a = np.frombuffer(plain, ev.RESULT_DTYPE).reshape(n_samples, n_sites)
            b = np.frombuffer(got, ev.RESULT_DTYPE).reshape(n_sites, n_samples)
            assert a.T.tobytes() == b.tobytes()
    finally:
        lib.svt_debug_coop(DEFAULT_MAX, 0)
