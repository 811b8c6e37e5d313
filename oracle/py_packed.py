"""TEST INFRASTRUCTURE -- CPU decoder of packed evidence (include/svtyper_hip.h: svt_packed_evidence).

An independent, plain-Python reading of the slot formats documented in svtyper_amd/csrc/svt_entry_formats.h:
slots -> the five tallies of every unit (after the zeroing rules), with the arithmetic of the reference
(svtyper/classic.py:296-435, singlesample.py:246-404; p_concordant parsers.py:861-882) restated in
oracle/py_oracle.py.  tests/test_packed_evidence.py compares it with the oracle on the canonical records the slots
were packed from: the encoder (svt_pack_evidence) is then pinned without a GPU.  Never imported by the product.
"""
from __future__ import annotations

import numpy as np

from svtyper_amd import evidence as ev
from . import py_oracle as po


def _pair_weights(f3: int, p_conc: bool, is_del: bool):
    """the paired-end branches of classic.py:359-405 for one fragment: (w_alt, w_ref)"""
    alt, ra, rb = bool(f3 & 1), bool(f3 & 2), bool(f3 & 4)
    w_alt = 1.0 if (alt and not (is_del and p_conc)) else 0.0       # classic.py:359-377
    w_ref = 0.0
    if (ra or rb) and (not (ra and rb) or is_del):                   # classic.py:398-401
        w_ref = ((1 if ra else 0) + (1 if rb else 0)) * (1.0 if p_conc else 0.0) / 2   # classic.py:402-405
    return w_alt, w_ref


def tally_packed(slots: np.ndarray, slot_offset: np.ndarray, units: np.ndarray, lib, common_mapq: int,
                 sso: bool) -> np.ndarray:
    """float64 [n_units, 5] tallies in ev.TALLY_NAMES order, zeroing rules applied.
    lib: the batch's library, or the list of its libraries (several: library switches in the pair streams)."""
    libs = list(lib) if isinstance(lib, (list, tuple)) else [lib]
    Ls = [po._Lib(x) for x in libs]
    out = np.zeros((len(units), 5))
    for u in range(len(units)):
        is_del = int(units["svtype"][u]) == 0
        var_length = int(units["var_length"][u])
        cur = 0                                  # every unit's pair stream starts in the context of library 0
        o0, o1, o2, o3 = (int(x) for x in slot_offset[3 * u:3 * u + 4])
        ref_seq = alt_seq = alt_clip = ref_span = alt_span = 0
        l_ref = l_seq = l_clip = 0

        # ---- stream 0: pair entries, eight half-words per slot
        half = slots[o0:o1].view(np.uint16).reshape(-1).tolist()
        k = 0
        while k < len(half):
            h = half[k]
            mq = common_mapq
            if h & 0x8000:                       # wide entry: the next half-word holds its two MAPQs
                assert k % 2 == 0, "wide entries start on a 4-byte boundary"
                mq = half[k + 1]
                k += 1
            k += 1
            f3, code = h & 7, (h >> 3) & 0xfff
            if f3 == 0:
                if h != 0:                       # library switch (l + 1) << 3: the entries behind it belong to library l
                    assert not (h & 0x8000) and len(libs) > 1 and code - 1 < len(libs), "library switch out of place"
                    cur = code - 1
                continue                         # (zero: a no-op half-word, padding)
            L, key_min, n_bins = Ls[cur], int(libs[cur].key_min), int(len(libs[cur].hist))
            off2 = min(var_length, n_bins)
            # code -> the two histogram keys: bins[code] and, for a DEL, bins[code - min(var_length, n_bins)];
            # codes >= n_bins name the second window only / neither (svt_entry_formats.h)
            k1 = key_min + code if code < n_bins else None
            i2 = code - off2 if is_del else None
            k2 = key_min + i2 if (i2 is not None and 0 <= i2 < n_bins) else None
            d1 = L.density(k1) if k1 is not None else 0
            d2 = L.density(k2) if k2 is not None else 0
            try:
                p_conc = (float(d1) * 0.95 / (0.95 * d1 + 0.05 * d2)) > 0.5     # parsers.py:876-882
            except ZeroDivisionError:
                p_conc = False
            w_alt, w_ref = _pair_weights(f3, p_conc, is_del)
            pp = po.prob_mapq(mq & 0xff) * po.prob_mapq(mq >> 8)
            alt_span += pp * w_alt
            ref_span += pp * w_ref

        # ---- streams 1, 2: seven MAPQ pairs per slot, flag bits in bytes 14, 15
        def weight_slots(lo, hi):
            for s in slots[lo:hi]:
                b = s.view(np.uint8).tolist()
                for e in range(7):
                    x, y = b[2 * e], b[2 * e + 1]
                    if x == 0 and y == 0:
                        continue                 # empty entry of the last slot
                    yield x, y, bool(b[14] >> e & 1), bool(b[15] >> e & 1)

        for x, y, first, _ in weight_slots(o1, o2):              # reference reads -> ref_seq (classic.py:306-315)
            if sso:
                if first:
                    ref_seq += l_ref
                    l_ref = 0
                l_ref += po.prob_mapq(x)
                l_ref += po.prob_mapq(y)
            else:
                ref_seq += po.prob_mapq(x)
                ref_seq += po.prob_mapq(y)
        for x, y, first, clip in weight_slots(o2, o3):           # split / clip candidates (classic.py:317-328)
            p = (po.prob_mapq(x) + po.prob_mapq(y)) / 2.0
            if sso:
                if clip:
                    if first:
                        alt_clip += l_clip
                        l_clip = 0
                    l_clip += p
                else:
                    if first:
                        alt_seq += l_seq
                        l_seq = 0
                    l_seq += p
            elif clip:
                alt_clip += p
            else:
                alt_seq += p
        if sso:
            ref_seq += l_ref
            alt_seq += l_seq
            alt_clip += l_clip
        # zeroing rules (classic.py:425-435)
        if (alt_seq + alt_clip) < 0.5 and alt_span >= 1:
            alt_seq = alt_clip = ref_seq = 0
        if alt_span < 0.5 and (alt_seq + alt_clip) >= 1:
            alt_span = ref_span = 0
        if alt_span + alt_seq == 0 and alt_clip > 0:
            alt_clip = 0
        out[u] = (ref_seq, alt_seq, alt_clip, ref_span, alt_span)
    return out
