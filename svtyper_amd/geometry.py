"""Fragment summaries for the device geometry stage (include/svtyper_hip.h: svt_fragment).

With `geometry="device"` the host no longer asks the breakpoint-dependent questions
(`is_ref_seq`, `is_pair_straddle`, `is_split_straddle`; svtyper/parsers.py:801-857,1122-1215): it
only condenses every read-fragment into a fixed-size, breakpoint-independent summary -- read
coordinates, the gap-free aligned intervals of each primary read, the two pieces of each valid
split candidate -- and `svt_geometry_kernel` derives the evidence records from (summary,
breakpoint) on the GPU.  What stays on the host is what needs the BAM record itself: CIGAR / SA
parsing and the split-candidate QC of `SplitRead.is_valid` (fragments.py).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import evidence as ev
from .evidence import CLibrary, LibraryTable

READ_PRESENT = 1
READ_REVERSE = 2
FRAG_PAIR = 1
FRAG_CONTINUATION = 2
BP_REV_A, BP_REV_B, BP_SKIP = 1, 2, 4

READ_FIELDS = [("tid", "<i4"), ("start", "<i4"), ("end", "<i4"), ("iv_start", "<i4", (2,)), ("iv_end", "<i4", (2,)),
               ("mapq", "u1"), ("flags", "u1"), ("reserved", "<u2")]
PIECE_FIELDS = [("tid", "<i4"), ("start", "<i4"), ("end", "<i4"), ("mapq", "u1"), ("flags", "u1"), ("reserved", "<u2")]
READ_DTYPE = np.dtype(READ_FIELDS)
PIECE_DTYPE = np.dtype(PIECE_FIELDS)
FRAGMENT_DTYPE = np.dtype([("read", READ_DTYPE, (2,)), ("seq", PIECE_DTYPE, (2,)), ("clip", PIECE_DTYPE, (2,))])
BREAKPOINT_DTYPE = np.dtype([("tid_a", "<i4"), ("pos_a", "<i4"), ("ci_a", "<i4", (2,)), ("tid_b", "<i4"),
                             ("pos_b", "<i4"), ("ci_b", "<i4", (2,)), ("var_length", "<i4"), ("sample", "<u2"),
                             ("svtype", "u1"), ("flags", "u1"), ("reserved", "<u4", (2,))])
assert READ_DTYPE.itemsize == 32 and PIECE_DTYPE.itemsize == 16
assert FRAGMENT_DTYPE.itemsize == 128 and BREAKPOINT_DTYPE.itemsize == 48

_ALIGNED = (True, False, False, False, False, False, False, True, True)   # M = X
_GAP = (False, False, True, True, False, False, False, False, False)     # D N break an interval
_I32 = 2**31 - 1


def aligned_intervals(read) -> List[List[int]]:
    """Maximal reference intervals a read covers with M/=/X bases and no D/N gap in between: the
    get_overlap() of svtyper/parsers.py:813 over a 2 m window equals 2 m exactly when the window lies
    inside one of them."""
    out: List[List[int]] = []
    p = read.reference_start
    open_iv = None
    for op, n in read.cigar:
        if _ALIGNED[op]:
            if open_iv is None:
                open_iv = [p, p + n]
            else:
                open_iv[1] = p + n
            p += n
        elif _GAP[op]:
            if open_iv is not None:
                out.append(open_iv)
                open_iv = None
            p += n
    if open_iv is not None:
        out.append(open_iv)
    return out


def _clip32(x) -> int:
    return int(min(max(int(x), -_I32 - 1), _I32))


def _mapq8(q) -> int:
    q = int(q)
    if q < 0:
        raise ValueError("MAPQ %d does not fit the fragment summary (0..255)" % q)
    return min(q, 255)      # prob_mapq(q) is exactly 1.0 from q = 163 on (packer._mapq)


_ABSENT_READ = [-1, 0, 0, 0, 0, 0, 0, 0]   # 8 words of a svt_read_summary
_ABSENT_PIECE = [0, 0, 0, 0]                # 4 words of a svt_piece_summary


def _read_words(read, tid_of, near: Sequence[int]) -> List[int]:
    ivs = aligned_intervals(read)
    if len(ivs) > 2:   # keep the two intervals closest to the unit's breakends (only they can contain a window)
        ivs.sort(key=lambda iv: min(0 if iv[0] <= q <= iv[1] else min(abs(iv[0] - q), abs(iv[1] - q)) for q in near))
        ivs = ivs[:2]
    while len(ivs) < 2:
        ivs.append((0, 0))
    packed = _mapq8(read.mapping_quality) | ((READ_PRESENT | (READ_REVERSE if read.is_reverse else 0)) << 8)
    return [tid_of(read.reference_name), read.reference_start, read.reference_end, ivs[0][0], ivs[1][0], ivs[0][1],
            ivs[1][1], packed]


def _piece_words(piece, tid_of) -> List[int]:
    packed = _mapq8(piece.mapping_quality) | ((READ_PRESENT | (READ_REVERSE if piece.is_reverse else 0)) << 8)
    return [-2 if piece.chrom is None else tid_of(piece.chrom), _clip32(piece.reference_start),
            _clip32(piece.reference_end), packed]


def summarise_fragments(fragments: Dict[str, object], breakpoint: dict, lib_index: Dict[int, int], tid_of) -> np.ndarray:
    """FRAGMENT_DTYPE array of one unit, in `sorted(query_name)` order (svtyper/classic.py:296).  A
    fragment with more than two primaries or two split candidates of one kind spills into continuation
    summaries, mirroring the records packer.pack_fragments would emit."""
    near = (breakpoint["A"]["pos"], breakpoint["B"]["pos"])
    rows: List[List[int]] = []
    for name in sorted(fragments.keys()):
        frag = fragments[name]
        lib = lib_index[id(frag.lib)]
        primaries = frag.primary_reads
        seq = [s for s in frag.split_reads if not s.is_soft_clip]
        clip = [s for s in frag.split_reads if s.is_soft_clip]
        n_rec = max(1, (len(primaries) + 1) // 2, len(seq), len(clip))
        for k in range(n_rec):
            ra = _read_words(primaries[2 * k], tid_of, near) if 2 * k < len(primaries) else list(_ABSENT_READ)
            rb = _read_words(primaries[2 * k + 1], tid_of, near) if 2 * k + 1 < len(primaries) else list(_ABSENT_READ)
            ra[7] |= lib << 16                                     # read[0].reserved: library index
            bits = (FRAG_PAIR if (k == 0 and frag.num_primary == 2) else 0) | (FRAG_CONTINUATION if k > 0 else 0)
            rb[7] |= bits << 16                                    # read[1].reserved: fragment bits
            row = ra + rb
            for cand in (seq, clip):
                if k < len(cand):
                    row += _piece_words(cand[k].query_left, tid_of) + _piece_words(cand[k].query_right, tid_of)
                else:
                    row += _ABSENT_PIECE + _ABSENT_PIECE
            rows.append(row)
    if not rows:
        return np.zeros(0, FRAGMENT_DTYPE)
    words = np.asarray(rows, dtype=np.int64).astype(np.uint32)    # 32 little-endian words per summary
    return np.ascontiguousarray(words).view(FRAGMENT_DTYPE).reshape(-1)


def breakpoint_record(breakpoint: dict, tid_of, sample_index: int = 0, skip: bool = False, libs: int = 0) -> np.ndarray:
    b = np.zeros(1, BREAKPOINT_DTYPE)
    b["reserved"][0, 0] = libs       # evidence.unit_libs(first, count) of the unit's sample (svt_unit.libs)
    A, B = breakpoint["A"], breakpoint["B"]
    b["tid_a"], b["pos_a"], b["ci_a"] = tid_of(A["chrom"]), _clip32(A["pos"]), [_clip32(x) for x in A["ci"]]
    b["tid_b"], b["pos_b"], b["ci_b"] = tid_of(B["chrom"]), _clip32(B["pos"]), [_clip32(x) for x in B["ci"]]
    b["svtype"] = ev.SVTYPE_CODE[breakpoint["svtype"]]
    if breakpoint["svtype"] == "DEL":
        b["var_length"] = _clip32(breakpoint["var_length"])
    b["sample"] = sample_index
    b["flags"] = (BP_REV_A if A["is_reverse"] else 0) | (BP_REV_B if B["is_reverse"] else 0) | (BP_SKIP if skip else 0)
    return b


class CFragmentBatch(C.Structure):
    _fields_ = [("n_units", C.c_uint64), ("frag_offset", C.POINTER(C.c_uint64)), ("breakpoints", C.c_void_p),
                ("fragments", C.c_void_p), ("n_libs", C.c_uint32), ("libs", C.POINTER(CLibrary)),
                ("split_weight", C.c_double), ("disc_weight", C.c_double), ("min_aligned", C.c_int32),
                ("split_slop", C.c_int32)]


@dataclass
class FragmentBatch:
    """CSR batch of fragment summaries (include/svtyper_hip.h: svt_fragment_batch)."""
    frag_offset: np.ndarray
    breakpoints: np.ndarray
    fragments: np.ndarray
    libs: List[LibraryTable]
    split_weight: float = 1.0
    disc_weight: float = 1.0
    min_aligned: int = 20
    split_slop: int = 3
    _keep: list = field(default_factory=list, repr=False)

    def __post_init__(self):
        self.frag_offset = np.ascontiguousarray(self.frag_offset, dtype=np.uint64)
        self.breakpoints = np.ascontiguousarray(self.breakpoints, dtype=BREAKPOINT_DTYPE)
        self.fragments = np.ascontiguousarray(self.fragments, dtype=FRAGMENT_DTYPE)

    @property
    def n_units(self) -> int:
        return int(self.breakpoints.shape[0])

    @property
    def n_fragments(self) -> int:
        return int(self.fragments.shape[0])

    def as_c(self) -> CFragmentBatch:
        clibs = (CLibrary * max(1, len(self.libs)))()
        keep = []
        for i, lib in enumerate(self.libs):
            h = np.ascontiguousarray(lib.hist, dtype=np.uint32)
            keep.append(h)
            clibs[i].hist = h.ctypes.data_as(C.POINTER(C.c_uint32))
            clibs[i].key_min = int(lib.key_min)
            clibs[i].n_bins = int(h.shape[0])
            clibs[i].mean = float(lib.mean)
            clibs[i].sd = float(lib.sd)
        cb = CFragmentBatch()
        cb.n_units = self.n_units
        cb.frag_offset = self.frag_offset.ctypes.data_as(C.POINTER(C.c_uint64))
        cb.breakpoints = self.breakpoints.ctypes.data
        cb.fragments = self.fragments.ctypes.data
        cb.n_libs = len(self.libs)
        cb.libs = clibs
        cb.split_weight = float(self.split_weight)
        cb.disc_weight = float(self.disc_weight)
        cb.min_aligned = int(self.min_aligned)
        cb.split_slop = int(self.split_slop)
        self._keep = [clibs, keep]
        return cb


class FragmentBatchBuilder:
    def __init__(self, libs: Sequence[LibraryTable], split_weight=1.0, disc_weight=1.0, min_aligned=20, split_slop=3):
        self.libs = list(libs)
        self.params = (float(split_weight), float(disc_weight), int(min_aligned), int(split_slop))
        self._bps: List[np.ndarray] = []
        self._frags: List[np.ndarray] = []
        self._off: List[int] = [0]

    def __len__(self):
        return len(self._bps)

    def add(self, bp_record: np.ndarray, fragments: Optional[np.ndarray]) -> int:
        if fragments is None:
            fragments = np.zeros(0, FRAGMENT_DTYPE)
        self._bps.append(bp_record)
        self._frags.append(fragments)
        self._off.append(self._off[-1] + int(fragments.shape[0]))
        return len(self._bps) - 1

    def build(self) -> FragmentBatch:
        bps = np.concatenate(self._bps) if self._bps else np.zeros(0, BREAKPOINT_DTYPE)
        fr = np.concatenate(self._frags) if self._frags else np.zeros(0, FRAGMENT_DTYPE)
        sw, dw, m, slop = self.params
        return FragmentBatch(np.asarray(self._off, np.uint64), bps, fr, self.libs, sw, dw, m, slop)
