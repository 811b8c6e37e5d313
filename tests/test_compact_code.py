"""The algebra behind the pair entries of packed evidence (svtyper_amd/csrc/svt_entry_formats.h: pair_code; consumed by
short_pair_dword in svt_unit_math.h): one table code per entry from which the kernel gets BOTH histogram indices
by clamping.  Exhaustive over small geometries, against the direct definition of the two look-ups
(svtyper/parsers.py:870-878).  Pure arithmetic -- the device code itself is covered by the gpu tests
(tests/test_hip_parity.py::test_histogram_windows, tests/test_packed_evidence.py)."""
import itertools

M32 = (1 << 32) - 1


def pair_code(ospan_len, key_min, n_bins, is_del, var_length):
    """mirror of the device function"""
    r = ospan_len - key_min
    in1 = 0 <= r < n_bins
    far = 2 * n_bins
    if not is_del:
        return r if in1 else far
    r2 = r - var_length
    in2 = 0 <= r2 < n_bins
    if var_length < n_bins:
        return r if 0 <= r < var_length + n_bins else far
    return r if in1 else (n_bins + r2 if in2 else far)


def kernel_indices(code, n_bins, is_del, var_length):
    """what the consumer computes (in bins; the kernel works in byte offsets, unsigned 32-bit)"""
    off2 = min(var_length, n_bins) if is_del else 0x80000000 // 8
    i1 = min(code, n_bins)
    i2 = min((code - off2) & M32, n_bins)
    return i1, i2


def direct_indices(ospan_len, key_min, n_bins, is_del, var_length):
    """hist[o] and hist[o - var_length] of parsers.py:870-878, out-of-range -> the sentinel bin n_bins"""
    r = ospan_len - key_min
    i1 = r if 0 <= r < n_bins else n_bins
    r2 = r - var_length
    i2 = r2 if (is_del and 0 <= r2 < n_bins) else n_bins
    return i1, i2


def test_code_gives_both_table_indices():
    checked = 0
    for n_bins, key_min in itertools.product((1, 2, 3, 7, 16, 33), (0, 2, 5)):
        for is_del in (False, True):
            for var_length in ((0,) if not is_del else (0, 1, 2, n_bins - 1, n_bins, n_bins + 1, 2 * n_bins, 5 * n_bins + 3, 1000)):
                if var_length < 0:
                    continue
                for ospan_len in range(0, key_min + max(var_length, 0) + 2 * n_bins + 8):
                    code = pair_code(ospan_len, key_min, n_bins, is_del, var_length)
                    assert 0 <= code <= 2 * n_bins                      # fits the 13-bit field for n_bins <= 4095
                    got = kernel_indices(code, n_bins, is_del, var_length)
                    want = direct_indices(ospan_len, key_min, n_bins, is_del, var_length)
                    assert got == want, (n_bins, key_min, is_del, var_length, ospan_len, code)
                    checked += 1
    assert checked > 10000


def test_far_spans_take_the_sentinel():
    for ospan_len in (0, 1, 10**6, 2**31 - 1):
        code = pair_code(ospan_len, 100, 50, True, 10**5)
        assert kernel_indices(code, 50, True, 10**5) == direct_indices(ospan_len, 100, 50, True, 10**5)


def test_segmented_batch_is_the_joined_batch():
    """evidence.SegmentedBatch (what a joint run hands to svt_batch_create_segments): the record array is the concatenation of the
    segments; lengths that do not add up to rec_offset[-1] are refused on the host already"""
    import numpy as np
    import pytest
    from svtyper_amd import evidence as ev, synth
    b = synth.make_multisample(40, 3, seed=5, mean_frags=6, sd_frags=3, min_frags=0, max_frags=15)
    cuts = [0, 7, 7, b.n_records // 2, b.n_records]
    seg = ev.SegmentedBatch(b.rec_offset, b.units, [b.records[x:y] for x, y in zip(cuts[:-1], cuts[1:])], b.libs, b.split_weight, b.disc_weight)
    assert seg.n_units == b.n_units and seg.n_records == b.n_records and len(seg.segments) == 4
    j = seg.joined()
    assert j.records.tobytes() == b.records.tobytes() and j.rec_offset.tobytes() == b.rec_offset.tobytes() and j.units.tobytes() == b.units.tobytes()
    with pytest.raises(ValueError):
        ev.SegmentedBatch(b.rec_offset, b.units, [b.records[:-1]], b.libs)
    with pytest.raises(ValueError):
        ev.SegmentedBatch(b.rec_offset[:-1], b.units, [b.records], b.libs)
