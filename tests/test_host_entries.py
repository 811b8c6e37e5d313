"""Host-only entry points of libsvtyper_hip.so (no device needed): svt_results_host_sq, svt_shard_bounds."""
import math

import numpy as np

from svtyper_amd import evidence as ev


def test_host_sq_is_the_reference_arithmetic():
    """SQ = abs(-10 * (GL[0] - log10(sum(10 ** GL)))) with the host libm (classic.py:473-481), for called units only."""
    from svtyper_amd import hip
    rng = np.random.default_rng(5)
    n = 4000
    rec = np.zeros(n, ev.RESULT_DTYPE)
    gl = -np.abs(rng.normal(0, 40, (n, 3)))
    gl[rng.integers(0, n, n // 10), rng.integers(0, 3, n // 10)] = 0.0
    gl[::97] = [-300.0, -305.5, -321.0]                      # deep in the subnormal band of 10 ** GL
    rec["gl"] = gl
    rec["gt"] = rng.choice([0, 1, 2, ev.GT_BLANK, ev.GT_SKIPPED, ev.GT_MISSING], n, p=[0.3, 0.3, 0.3, 0.04, 0.03, 0.03])
    rec["sq"] = 12345.0                                      # must be overwritten for called units, kept otherwise
    res = hip.host_sq(ev.Results(rec.copy()))
    for k in range(n):
        if rec["gt"][k] >= 0:
            s = sum(10.0 ** float(x) for x in gl[k])
            want = abs(-10.0 * (float(gl[k, 0]) - math.log(s) / math.log(10.0))) if s > 0.0 else None
            if want is not None:
                assert float(res.sq[k]).hex() == float(want).hex(), (k, gl[k])
        else:
            assert res.sq[k] == 12345.0


def test_shard_bounds_balance_bytes_and_keep_groups_whole():
    from svtyper_amd import hip
    rng = np.random.default_rng(7)
    counts = rng.integers(0, 400, 12_800)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    for shards, group in ((1, 1), (3, 1), (8, 32), (5, 7)):
        b = hip.shard_bounds(off, shards, group)
        assert len(b) == shards and b[0][0] == 0 and b[-1][1] == len(counts)
        assert all(lo <= hi for lo, hi in b) and all(b[i][1] == b[i + 1][0] for i in range(shards - 1))
        assert all(lo % group == 0 for lo, _ in b)
        cost = [int(16 * (off[hi] - off[lo]) + 112 * (hi - lo)) for lo, hi in b]
        biggest_group = 16 * int(counts.max()) * group + 112 * group
        assert max(cost) - min(cost) <= 2 * biggest_group or shards == 1


def test_library_hint_that_does_not_fit_is_no_hint():
    """SVT_UNIT_LIBS (ABI 18): first < 65536 as low byte | count << 8 | high byte << 16 -- the word ABI <= 17 wrote for
    first < 256 --, count <= 255; 256 libraries in one sample, or a first library beyond 65535, must become `no hint` (0),
    never a value with reserved bits (24..31) set."""
    assert ev.unit_libs(3, 2) == (3 | 2 << 8)
    assert ev.unit_libs(0, 255) == (255 << 8)
    assert ev.unit_libs(256, 1) == (1 << 8 | 1 << 16) and ev.unit_libs(0x1234, 7) == (0x34 | 7 << 8 | 0x12 << 16)
    assert ev.unit_libs(0, 256) == 0 and ev.unit_libs(65536, 1) == 0 and ev.unit_libs(5, 0) == 0
    for first in list(range(0, 300, 37)) + [255, 256, 4095, 65535, 65536, 70000]:
        for count in (0, 1, 3, 255, 256, 300):
            h = ev.unit_libs(first, count)
            assert h >> 24 == 0
            if h:
                assert (ev.unit_libs_first(h), ev.unit_libs_count(h)) == (first, count)


def test_pinned_results_memory_lives_as_long_as_any_view():
    """hip.pinned_results: the page-locked block belongs to the array (not to the Results wrapper), so code that keeps
    only `.rec`, or a slice of it, keeps the memory."""
    import gc
    from svtyper_amd import hip
    r = hip.pinned_results(100)
    rec = r.rec
    tail = rec[50:]
    del r
    gc.collect()
    rec["sq"] = 1.5             # (the block comes from a pool: its old contents are whatever they were)
    rec["gt"] = 0
    tail["gt"] = 2
    assert float(rec["sq"].sum()) == 150.0 and int(rec["gt"].sum()) == 100
    base = tail
    while getattr(base, "base", None) is not None:
        base = base.base
    assert hasattr(base, "_owner")          # the ctypes array that owns the pooled block
    del rec, base
    gc.collect()
    assert int(tail["gt"].sum()) == 100      # still alive through the slice


def test_out_buffers_are_checked_before_the_c_side_writes_through_them():
    import pytest
    from svtyper_amd import hip
    small = hip.Results.empty(3)
    with pytest.raises(ValueError):
        hip._check_out(small, 4)
    strided = hip.Results.empty(8)
    strided.rec = strided.rec[::2]
    with pytest.raises(ValueError):
        hip._check_out(strided, 4)
    assert hip._check_out(hip.Results.empty(4), 4).n_units == 4
