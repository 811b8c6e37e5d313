"""Packed read-evidence batches: the data that crosses the C ABI (include/svtyper_hip.h).

The host side of the hot path turns every read-fragment the reference would visit in
its per-variant loop (svtyper/classic.py:296-408, svtyper/singlesample.py:246-353)
into one 16-byte ``svt_record``; a (breakpoint, sample) *unit* owns a contiguous,
sorted-by-query-name run of records (CSR).  This module holds the numpy mirrors of the
C structs, the batch container and the ctypes views handed to the library.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

# --------------------------------------------------------------------------- constants
SVTYPE_CODE = {"DEL": 0, "DUP": 1, "INV": 2, "BND": 3}  # classic.py:228
SVTYPE_NAME = {v: k for k, v in SVTYPE_CODE.items()}

FLAG_SSO_ASSOCIATION = 0x1
FLAG_GENERAL_TABLES = 0x10     # keep every table in L2 (the general mode): measurements, table-path comparisons
FLAG_RESULT96 = 0x20           # 96-byte result records on the device (svt_result96); the host API still returns 128-byte records

REC_ALT_STRADDLE = 1 << 0
REC_REF_STRADDLE_A = 1 << 1
REC_REF_STRADDLE_B = 1 << 2
REC_CONTINUATION = 1 << 3
REC_HAS_PAIR = 1 << 4
REC_LIB_SHIFT = 8
REC_FLAG_MASK = 0x0000FF1F

UNIT_SKIP = 1 << 0

GT_HOMREF, GT_HET, GT_HOMALT = 0, 1, 2
GT_MISSING, GT_BLANK, GT_SKIPPED = -1, -2, -3
GT_STRING = {0: "0/0", 1: "0/1", 2: "1/1", -1: "./.", -2: "./.", -3: "./."}

COUNT_NAMES = ("QR", "QA", "GQ", "DP", "RO", "AO", "RS", "AS", "ASC", "RP", "AP")
TALLY_NAMES = ("ref_seq", "alt_seq", "alt_clip", "ref_span", "alt_span")
N_COUNTS = len(COUNT_NAMES)
N_TALLIES = len(TALLY_NAMES)

# --------------------------------------------------------------------------- dtypes
RECORD_DTYPE = np.dtype(
    [
        ("ospan_len", "<i4"),
        ("mapq_a", "u1"),
        ("mapq_b", "u1"),
        ("rs_a", "u1"),
        ("rs_b", "u1"),
        ("seq_l", "u1"),
        ("seq_r", "u1"),
        ("clip_l", "u1"),
        ("clip_r", "u1"),
        ("flags", "<u4"),
    ],
    align=False,
)
assert RECORD_DTYPE.itemsize == 16

UNIT_DTYPE = np.dtype(
    [
        ("var_length", "<i4"),
        ("pos_delta", "<i4"),
        ("sample", "<u2"),
        ("svtype", "u1"),
        ("flags", "u1"),
        ("libs", "<u4"),      # optional hint: first library | library count << 8 of the unit's sample (0 = none)
    ],
    align=False,
)
assert UNIT_DTYPE.itemsize == 16


def unit_libs(first: int, count: int) -> int:
    """SVT_UNIT_LIBS(first, count): the libraries of a unit's sample are libs[first .. first + count) of the batch.
    first < 65536 (its low byte | count << 8 | its high byte << 16: ABI 18, the same word as before for first < 256),
    count <= 255: a sample that cannot be described (no library, more than 255 libraries, a first index beyond 65535)
    gets 0 = "no hint" -- the pass then takes such a batch without windows instead of rejecting it."""
    first, count = int(first), int(count)
    if count <= 0 or count > 255 or first < 0 or first > 65535:
        return 0
    return (first & 0xFF) | (count << 8) | ((first >> 8) << 16)


def unit_libs_first(hint):
    """SVT_UNIT_LIBS_FIRST (scalars or numpy arrays)"""
    return (hint & 0xFF) | (((hint >> 16) & 0xFF) << 8)


def unit_libs_count(hint):
    """SVT_UNIT_LIBS_COUNT"""
    return (hint >> 8) & 0xFF

RESULT_DTYPE = np.dtype(
    [
        ("gl", "<f8", (3,)),
        ("sq", "<f8"),
        ("tallies", "<f8", (N_TALLIES,)),
        ("counts", "<i4", (N_COUNTS,)),
        ("gt", "i1"),
        ("pad", "u1", (11,)),
    ],
    align=False,
)
assert RESULT_DTYPE.itemsize == 128
# the device record under FLAG_RESULT96 (svt_result96): svt_result without the counts that follow from the tallies
# + the index of the unit it belongs to: the records lie in the order the kernel finishes them (whole cache lines per wave)
RESULT96_DTYPE = np.dtype([("gl", "<f8", (3,)), ("sq", "<f8"), ("tallies", "<f8", (5,)), ("qr", "<i4"), ("qa", "<i4"), ("gq", "<i4"),
                           ("gt", "i1"), ("pad", "u1", (3,)), ("unit", "<u4"), ("pad2", "<u4")])
assert RESULT96_DTYPE.itemsize == 96
NO_UNIT = 0xFFFFFFFF           # svt_result96.unit of a padding record


# --------------------------------------------------------------------------- ctypes structs
class CLibrary(C.Structure):
    _fields_ = [
        ("hist", C.POINTER(C.c_uint32)),
        ("key_min", C.c_int32),
        ("n_bins", C.c_uint32),
        ("mean", C.c_double),
        ("sd", C.c_double),
    ]


class CEvidenceBatch(C.Structure):
    _fields_ = [
        ("n_units", C.c_uint64),
        ("rec_offset", C.POINTER(C.c_uint64)),
        ("units", C.c_void_p),
        ("records", C.c_void_p),
        ("n_libs", C.c_uint32),
        ("libs", C.POINTER(CLibrary)),
        ("split_weight", C.c_double),
        ("disc_weight", C.c_double),
    ]


class CPackedEvidence(C.Structure):
    """include/svtyper_hip.h: svt_packed_evidence"""
    _fields_ = [
        ("n_units", C.c_uint64),
        ("n_slots", C.c_uint64),
        ("n_records", C.c_uint64),
        ("slot_offset", C.POINTER(C.c_uint32)),
        ("units", C.c_void_p),
        ("slots", C.c_void_p),
        ("common_mapq", C.c_uint32),
        ("n_libs", C.c_uint32),
        ("libs", C.POINTER(CLibrary)),
        ("split_weight", C.c_double),
        ("disc_weight", C.c_double),
    ]


# --------------------------------------------------------------------------- library table
@dataclass
class LibraryTable:
    """Dense insert-size histogram of one library (svtyper/parsers.py:406-430).

    ``hist[k - key_min]`` is ``Library.hist[k]``; ``mean``/``sd`` are the library
    moments used for the small-DEL gate (classic.py:339) and the non-DEL
    ``var_length`` default (parsers.py:874-875)."""

    hist: np.ndarray
    key_min: int
    mean: float
    sd: float
    name: str = ""

    @classmethod
    def from_counter(cls, hist: dict, mean: float, sd: float, name: str = "") -> "LibraryTable":
        keys = [int(k) for k in hist.keys()]
        if not keys:
            return cls(np.zeros(1, np.uint32), 0, float(mean), float(sd), name)
        kmin, kmax = min(keys), max(keys)
        dense = np.zeros(kmax - kmin + 1, dtype=np.uint32)
        for k, v in hist.items():
            dense[int(k) - kmin] = int(v)
        return cls(dense, kmin, float(mean), float(sd), name)

    @property
    def n_total(self) -> int:
        return int(self.hist.sum(dtype=np.uint64))


# --------------------------------------------------------------------------- batch
@dataclass
class EvidenceBatch:
    """CSR batch of units (include/svtyper_hip.h: svt_evidence_batch)."""

    rec_offset: np.ndarray  # uint64 [n_units + 1]
    units: np.ndarray  # UNIT_DTYPE [n_units]
    records: np.ndarray  # RECORD_DTYPE [n_records]
    libs: List[LibraryTable]
    split_weight: float = 1.0
    disc_weight: float = 1.0
    _keep: list = field(default_factory=list, repr=False)

    def __post_init__(self):
        self.rec_offset = np.ascontiguousarray(self.rec_offset, dtype=np.uint64)
        self.units = np.ascontiguousarray(self.units, dtype=UNIT_DTYPE)
        self.records = np.ascontiguousarray(self.records, dtype=RECORD_DTYPE)
        if self.rec_offset.shape[0] != self.units.shape[0] + 1:
            raise ValueError("rec_offset must have n_units + 1 entries")
        if self.units.shape[0] and int(self.rec_offset[-1]) != self.records.shape[0]:
            raise ValueError("rec_offset[-1] must equal the number of records")

    @property
    def n_units(self) -> int:
        return int(self.units.shape[0])

    @property
    def n_records(self) -> int:
        return int(self.records.shape[0])

    def algorithmic_bytes(self) -> int:
        """sum_u (16 F(u) + 16 + 96)."""
        return 16 * self.n_records + (16 + 96) * self.n_units

    def slice(self, lo: int, hi: int) -> "EvidenceBatch":
        """Units [lo, hi) as an independent batch (used to shard across ranks)."""
        r0, r1 = int(self.rec_offset[lo]), int(self.rec_offset[hi])
        return EvidenceBatch(
            self.rec_offset[lo : hi + 1] - np.uint64(r0),
            self.units[lo:hi],
            self.records[r0:r1],
            self.libs,
            self.split_weight,
            self.disc_weight,
        )

    def as_c(self) -> CEvidenceBatch:
        """ctypes view; the returned struct borrows this object's buffers."""
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
        cb = CEvidenceBatch()
        cb.n_units = self.n_units
        cb.rec_offset = self.rec_offset.ctypes.data_as(C.POINTER(C.c_uint64))
        cb.units = self.units.ctypes.data
        cb.records = self.records.ctypes.data
        cb.n_libs = len(self.libs)
        cb.libs = clibs
        cb.split_weight = float(self.split_weight)
        cb.disc_weight = float(self.disc_weight)
        self._keep = [clibs, keep]
        return cb


class SegmentedBatch:
    """A CSR batch whose RECORD array is handed over in pieces (include/svtyper_hip.h: svt_batch_create_segments): the record
    array of the batch is the concatenation of `segments`.  What a joint run builds from one reader per sample -- units
    sample-major, every sample's records where its reader left them."""

    def __init__(self, rec_offset, units, segments, libs, split_weight: float = 1.0, disc_weight: float = 1.0):
        self.rec_offset = np.ascontiguousarray(rec_offset, dtype=np.uint64)
        self.units = np.ascontiguousarray(units, dtype=UNIT_DTYPE)
        self.segments = [np.ascontiguousarray(x, dtype=RECORD_DTYPE) for x in segments]
        self.libs = libs
        self.split_weight = split_weight
        self.disc_weight = disc_weight
        if self.rec_offset.shape[0] != self.units.shape[0] + 1:
            raise ValueError("rec_offset must have n_units + 1 entries")
        if self.units.shape[0] and int(self.rec_offset[-1]) != self.n_records:
            raise ValueError("rec_offset[-1] must equal the number of records of all segments")

    @property
    def n_units(self) -> int:
        return int(self.units.shape[0])

    @property
    def n_records(self) -> int:
        return sum(int(x.shape[0]) for x in self.segments)

    def joined(self) -> "EvidenceBatch":
        """The same batch with its records in one array (a copy of 16 bytes per record)."""
        recs = np.concatenate(self.segments) if self.segments else np.zeros(0, RECORD_DTYPE)
        return EvidenceBatch(self.rec_offset, self.units, recs, self.libs, self.split_weight, self.disc_weight)


class Results:
    """Result records (include/svtyper_hip.h: svt_result[n_units]), one 128-byte record per unit."""

    def __init__(self, rec: np.ndarray):
        self.rec = np.ascontiguousarray(rec, dtype=RESULT_DTYPE)
        self.site_qual = None   # float64 [n_sites] when the engine also accumulated QUAL over the samples

    @classmethod
    def empty(cls, n: int) -> "Results":
        return cls(np.zeros(n, RESULT_DTYPE))

    @property
    def n_units(self) -> int:
        return int(self.rec.shape[0])

    @property
    def gl(self) -> np.ndarray:  # [n, 3]
        return self.rec["gl"]

    @property
    def sq(self) -> np.ndarray:
        return self.rec["sq"]

    @property
    def tallies(self) -> np.ndarray:  # [n, 5] in TALLY_NAMES order
        return self.rec["tallies"]

    @property
    def counts(self) -> np.ndarray:  # [n, 11] in COUNT_NAMES order
        return self.rec["counts"]

    @property
    def gt(self) -> np.ndarray:
        return self.rec["gt"]

    def count(self, name: str) -> np.ndarray:
        return self.rec["counts"][:, COUNT_NAMES.index(name)]

    def tally(self, name: str) -> np.ndarray:
        return self.rec["tallies"][:, TALLY_NAMES.index(name)]

    def ptr(self) -> int:
        return int(self.rec.ctypes.data)


def concat_batches(batches: Sequence[EvidenceBatch]) -> EvidenceBatch:
    """Concatenate batches that share libraries and weights."""
    first = batches[0]
    offs = [np.zeros(1, np.uint64)]
    base = 0
    for b in batches:
        offs.append(b.rec_offset[1:] + np.uint64(base))
        base += b.n_records
    return EvidenceBatch(
        np.concatenate(offs),
        np.concatenate([b.units for b in batches]),
        np.concatenate([b.records for b in batches]),
        first.libs,
        first.split_weight,
        first.disc_weight,
    )
