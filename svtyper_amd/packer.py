"""Host-side evidence packer: read-fragments of one (breakpoint, sample) -> 16-byte records.

This is the seam between the geometry layer (fragments.py: which reads support what) and the
device (include/svtyper_hip.h: svt_record / svt_unit).  It walks the fragments exactly the way
the reference's per-variant loop does -- ``sorted(query_name)`` order, primaries before split
candidates (svtyper/classic.py:296-408, svtyper/singlesample.py:246-353) -- and asks the
fragment objects the reference's own yes/no questions (`is_ref_seq`, `is_split_straddle`,
`is_pair_straddle`).  It is duck-typed on purpose: tests/golden/make_golden.py feeds it the
*reference's* SamFragment objects to produce golden records, the product feeds it
svtyper_amd.fragments objects.

No arithmetic of the hot path happens here: the weights (prob_mapq), the insert-size test
(p_concordant), the small-deletion gate, the tallies and the likelihood are all evaluated on
the device.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import evidence as ev
from .evidence import EvidenceBatch, LibraryTable, RECORD_DTYPE, UNIT_DTYPE

_I32_MAX = 2**31 - 1
_I32_MIN = -(2**31)


def _mapq(x) -> int:
    """MAPQ as the record's byte.  A value above 255 (only an SA tag can carry one: BAM's own field is a byte) is
    stored as 255: prob_mapq(q) = 1 - 10**(-q / 10) is exactly 1.0 in binary64 for every q >= 163, so the weight the
    reference would compute is unchanged.  A negative one has no counterpart in the record."""
    q = int(x)
    if q < 0:
        raise ValueError("MAPQ %d does not fit the evidence record (0..255)" % q)
    return min(q, 255)


def _clamp32(x) -> int:
    return int(min(max(int(x), _I32_MIN), _I32_MAX))


def pack_fragments(fragments: Dict[str, object], breakpoint: dict, lib_index: Dict[int, int],
                   min_aligned: int, split_slop: int) -> np.ndarray:
    """Evidence records of one unit.

    fragments : {query_name: SamFragment-like}
    breakpoint: {'svtype', optional 'var_length', 'A': {chrom,pos,ci,is_reverse}, 'B': {...}}
                (svtyper/parsers.py:149-154,190-203; positions already carry the +1 of
                reverse-strand sides, classic.py:276-277)
    lib_index : id(library object) -> index into the batch's library tables
    """
    A, B = breakpoint["A"], breakpoint["B"]
    chromA, posA, ciA, o1 = A["chrom"], A["pos"], A["ci"], A["is_reverse"]
    chromB, posB, ciB, o2 = B["chrom"], B["pos"], B["ci"], B["is_reverse"]
    svtype = breakpoint["svtype"]
    zero_ci = [0, 0]
    rows: List[tuple] = []

    for name in sorted(fragments.keys()):                       # classic.py:296
        frag = fragments[name]
        lib = lib_index[id(frag.lib)]
        base_flags = lib << ev.REC_LIB_SHIFT

        # ---- gated MAPQs of the primary reads (classic.py:306-311)
        rs = []
        for read in frag.primary_reads:
            hit = (frag.is_ref_seq(read, None, chromA, posA, ciA, min_aligned)
                   or frag.is_ref_seq(read, None, chromB, posB, ciB, min_aligned))
            rs.append(_mapq(read.mapping_quality) if hit else 0)

        # ---- gated MAPQs of the split candidates (classic.py:317-328)
        seq, clip = [], []
        for split in frag.split_reads:
            left, right = split.is_split_straddle(chromA, posA, ciA, chromB, posB, ciB, o1, o2,
                                                  svtype, split_slop)
            pair = (_mapq(split.query_left.mapping_quality) if left else 0,
                    _mapq(split.query_right.mapping_quality) if right else 0)
            (clip if split.is_soft_clip else seq).append(pair)

        # ---- paired-end bits (classic.py:339-396), WITHOUT the small-deletion gate
        flags = base_flags
        mq_a = mq_b = ospan = 0
        n_primary = frag.num_primary
        if n_primary >= 1:
            mq_a = _mapq(frag.primary_reads[0].mapping_quality)
        if n_primary >= 2:
            mq_b = _mapq(frag.primary_reads[1].mapping_quality)
        if n_primary == 2:
            flags |= ev.REC_HAS_PAIR
            o = frag.get_ospan()
            ospan = abs(o[1] - o[0])                             # parsers.py:866-869
            alt = frag.is_pair_straddle(chromA, posA, ciA, chromB, posB, ciB, o1, o2, min_aligned, frag.lib)
            if not alt and svtype == "INV":                      # classic.py:349-357
                alt = frag.is_pair_straddle(chromA, posA, ciA, chromB, posB, ciB, not o1, not o2,
                                            min_aligned, frag.lib)
            if alt:
                flags |= ev.REC_ALT_STRADDLE
            if frag.is_pair_straddle(chromA, posA, zero_ci, chromA, posA, zero_ci, False, True,
                                     min_aligned, frag.lib):     # classic.py:387-391
                flags |= ev.REC_REF_STRADDLE_A
            if frag.is_pair_straddle(chromB, posB, zero_ci, chromB, posB, zero_ci, False, True,
                                     min_aligned, frag.lib):     # classic.py:392-396
                flags |= ev.REC_REF_STRADDLE_B
        else:
            # is_pair_straddle() is False unless exactly two primaries exist (parsers.py:827);
            # the pair product then never enters a tally
            mq_a = mq_b = 0

        # ---- emit: first record carries the pair; extra primaries / same-kind candidates go into
        # continuation records (only the sso association distinguishes them)
        n_rec = max(1, (len(rs) + 1) // 2, len(seq), len(clip))
        for k in range(n_rec):
            ra = rs[2 * k] if 2 * k < len(rs) else 0
            rb = rs[2 * k + 1] if 2 * k + 1 < len(rs) else 0
            sl, sr = seq[k] if k < len(seq) else (0, 0)
            cl, cr = clip[k] if k < len(clip) else (0, 0)
            if k == 0:
                rows.append((min(ospan, _I32_MAX), mq_a, mq_b, ra, rb, sl, sr, cl, cr, flags))
            else:
                rows.append((0, 0, 0, ra, rb, sl, sr, cl, cr, base_flags | ev.REC_CONTINUATION))

    rec = np.zeros(len(rows), RECORD_DTYPE)
    if rows:
        arr = np.array(rows, dtype=np.int64)
        for i, name in enumerate(RECORD_DTYPE.names):
            rec[name] = arr[:, i]
    return rec


def unit_header(breakpoint: dict, sample_index: int = 0, skip: bool = False, libs: int = 0) -> np.ndarray:
    """`libs`: evidence.unit_libs(first, count) of the unit's sample (svt_unit.libs), 0 = no hint."""
    u = np.zeros(1, UNIT_DTYPE)
    u["libs"] = libs
    svtype = breakpoint["svtype"]
    u["svtype"] = ev.SVTYPE_CODE[svtype]
    if svtype == "DEL":
        u["var_length"] = _clamp32(breakpoint["var_length"])     # classic.py:268
    u["pos_delta"] = _clamp32(breakpoint["B"]["pos"] - breakpoint["A"]["pos"])  # classic.py:339
    u["sample"] = sample_index
    u["flags"] = ev.UNIT_SKIP if skip else 0
    return u


class BatchBuilder:
    """Accumulates units into one EvidenceBatch (CSR)."""

    def __init__(self, libs: Sequence[LibraryTable], split_weight: float = 1.0, disc_weight: float = 1.0):
        self.libs = list(libs)
        self.split_weight = float(split_weight)
        self.disc_weight = float(disc_weight)
        self._units: List[np.ndarray] = []
        self._records: List[np.ndarray] = []
        self._offsets: List[int] = [0]

    def __len__(self):
        return len(self._units)

    def add(self, unit: np.ndarray, records: Optional[np.ndarray]) -> int:
        """Returns the unit's index in the batch."""
        if records is None:
            records = np.zeros(0, RECORD_DTYPE)
        self._units.append(unit)
        self._records.append(records)
        self._offsets.append(self._offsets[-1] + int(records.shape[0]))
        return len(self._units) - 1

    def build(self) -> EvidenceBatch:
        units = np.concatenate(self._units) if self._units else np.zeros(0, UNIT_DTYPE)
        recs = np.concatenate(self._records) if self._records else np.zeros(0, RECORD_DTYPE)
        return EvidenceBatch(np.asarray(self._offsets, np.uint64), units, recs, self.libs,
                             self.split_weight, self.disc_weight)
