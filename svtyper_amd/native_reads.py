"""ctypes binding of the native BAM reader + fragment summariser (include/svtyper_reads.h).

`NativeBam` offers the handful of `pysam.AlignmentFile` attributes the library / sample layer needs
(header['RG'], references, lengths, gettid) and `summarise()`, which fetches, assembles and
condenses the read-fragments of many (breakpoint, sample) units in C++ threads.  The summaries feed
the device geometry stage directly (`geometry="device"`), so with `reader="native"` no per-read
Python object is created at all.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import hip
from .geometry import BREAKPOINT_DTYPE, FRAGMENT_DTYPE

FETCH_DTYPE = np.dtype([("tid_a", "<i4"), ("lo_a", "<i4"), ("hi_a", "<i4"),
                        ("tid_b", "<i4"), ("lo_b", "<i4"), ("hi_b", "<i4")])
assert FETCH_DTYPE.itemsize == 24

COUNT_CLASSIC, COUNT_SSO = 0, 1


class _Args(C.Structure):
    _fields_ = [("n_units", C.c_uint64), ("windows", C.c_void_p), ("breakpoints", C.c_void_p),
                ("n_read_groups", C.c_uint32), ("read_groups", C.POINTER(C.c_char_p)),
                ("read_group_lib", C.POINTER(C.c_int32)), ("max_reads", C.c_int64), ("count_mode", C.c_int32),
                ("n_threads", C.c_int32)]


class _Summaries(C.Structure):
    _fields_ = [("frag_offset", C.POINTER(C.c_uint64)), ("fragments", C.c_void_p), ("skipped", C.POINTER(C.c_uint8))]


class _EvidenceParams(C.Structure):
    _fields_ = [("n_libs", C.c_uint32), ("lib_flank", C.POINTER(C.c_double)), ("min_aligned", C.c_int32), ("split_slop", C.c_int32)]


class _Evidence(C.Structure):
    _fields_ = [("rec_offset", C.POINTER(C.c_uint64)), ("records", C.c_void_p), ("skipped", C.POINTER(C.c_uint8))]


class _LibraryScan(C.Structure):
    _fields_ = [("read_length", C.c_int64), ("in_lib", C.c_uint64), ("total", C.c_uint64), ("n_hist", C.c_uint64),
                ("hist_keys", C.POINTER(C.c_int64)), ("hist_counts", C.POINTER(C.c_uint64))]


_declared = False


def _lib():
    global _declared
    L = hip.load()
    if not _declared:
        L.svt_bam_open.restype = C.c_int
        L.svt_bam_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.svt_bam_close.restype = None
        L.svt_bam_close.argtypes = [C.c_void_p]
        L.svt_bam_n_references.restype = C.c_int32
        L.svt_bam_n_references.argtypes = [C.c_void_p]
        L.svt_bam_reference_name.restype = C.c_char_p
        L.svt_bam_reference_name.argtypes = [C.c_void_p, C.c_int32]
        L.svt_bam_reference_length.restype = C.c_int64
        L.svt_bam_reference_length.argtypes = [C.c_void_p, C.c_int32]
        L.svt_bam_tid.restype = C.c_int32
        L.svt_bam_tid.argtypes = [C.c_void_p, C.c_char_p]
        L.svt_bam_header_text.restype = C.c_char_p
        L.svt_bam_header_text.argtypes = [C.c_void_p]
        L.svt_bam_summarise.restype = C.c_int
        L.svt_bam_summarise.argtypes = [C.c_void_p, C.POINTER(_Args), C.POINTER(_Summaries)]
        L.svt_summaries_free.restype = None
        L.svt_summaries_free.argtypes = [C.POINTER(_Summaries)]
        L.svt_bam_evidence.restype = C.c_int
        L.svt_bam_evidence.argtypes = [C.c_void_p, C.POINTER(_Args), C.POINTER(_EvidenceParams), C.POINTER(_Evidence)]
        L.svt_evidence_free.restype = None
        L.svt_evidence_free.argtypes = [C.POINTER(_Evidence)]
        L.svt_bam_scan_library.restype = C.c_int
        L.svt_bam_scan_library.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.c_int64, C.POINTER(_LibraryScan)]
        L.svt_library_scan_free.restype = None
        L.svt_library_scan_free.argtypes = [C.POINTER(_LibraryScan)]
        _declared = True
    return L


class NativeBam:
    """An indexed BAM opened by the C++ reader."""

    def __init__(self, path: str):
        L = _lib()
        self._L = L
        self._h = C.c_void_p()
        hip._check(L.svt_bam_open(path.encode(), C.byref(self._h)))
        self.filename = path
        n = L.svt_bam_n_references(self._h)
        self.references = tuple(L.svt_bam_reference_name(self._h, i).decode() for i in range(n))
        self.lengths = tuple(int(L.svt_bam_reference_length(self._h, i)) for i in range(n))
        self._tid = {r: i for i, r in enumerate(self.references)}
        text = (L.svt_bam_header_text(self._h) or b"").decode("ascii", "replace")
        self.header: Dict[str, list] = {}
        for line in text.splitlines():
            if not line.startswith("@") or line.startswith("@CO"):
                continue
            parts = line.split("\t")
            rec = {f[:2]: f[3:] for f in parts[1:] if len(f) >= 3 and f[2] == ":"}
            self.header.setdefault(parts[0][1:], []).append(rec)

    def gettid(self, name: str) -> int:
        return self._tid.get(name, -1)

    def close(self):
        if self._h:
            self._L.svt_bam_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scan_library(self, read_groups: Sequence[str], num_samp: int):
        """(read_length, {template_length: count}, reads of the library among the first 100 000, that total):
        the three scans of Library.from_bam (svtyper/parsers.py:501-576) in C++."""
        names = (C.c_char_p * max(1, len(read_groups)))(*[rg.encode() for rg in read_groups])
        out = _LibraryScan()
        hip._check(self._L.svt_bam_scan_library(self._h, len(read_groups), names, int(num_samp), C.byref(out)))
        try:
            n = int(out.n_hist)
            keys = np.ctypeslib.as_array(out.hist_keys, shape=(max(n, 1),))[:n].tolist()
            counts = np.ctypeslib.as_array(out.hist_counts, shape=(max(n, 1),))[:n].tolist()
            return int(out.read_length), dict(zip(keys, counts)), int(out.in_lib), int(out.total)
        finally:
            self._L.svt_library_scan_free(C.byref(out))

    def evidence(self, windows: np.ndarray, breakpoints: np.ndarray, read_groups: Sequence[str],
                 read_group_lib: Sequence[int], max_reads: Optional[int], count_mode: int, lib_flank: Sequence[float],
                 min_aligned: int, split_slop: int, n_threads: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """svt_bam_evidence: (rec_offset uint64 [n+1], records RECORD_DTYPE, skipped uint8 [n]) -- the units' 16-byte
        evidence records, the geometry predicates evaluated in the reader's threads (`lib_flank`: mean + 3 sd of every
        library of the batch).  What summarise() + the device geometry stage produce, byte for byte."""
        from .evidence import RECORD_DTYPE
        windows = np.ascontiguousarray(windows, dtype=FETCH_DTYPE)
        breakpoints = np.ascontiguousarray(breakpoints, dtype=BREAKPOINT_DTYPE)
        n = int(windows.shape[0])
        if breakpoints.shape[0] != n:
            raise ValueError("windows and breakpoints must have the same length")
        names = (C.c_char_p * max(1, len(read_groups)))(*[rg.encode() for rg in read_groups])
        libs = (C.c_int32 * max(1, len(read_groups)))(*[int(x) for x in read_group_lib])
        flank = (C.c_double * max(1, len(lib_flank)))(*[float(x) for x in lib_flank])
        a = _Args(n, windows.ctypes.data, breakpoints.ctypes.data, len(read_groups), names, libs,
                  -1 if max_reads is None else int(max_reads), int(count_mode), int(n_threads))
        g = _EvidenceParams(len(lib_flank), flank, int(min_aligned), int(split_slop))
        out = _Evidence()
        hip._check(self._L.svt_bam_evidence(self._h, C.byref(a), C.byref(g), C.byref(out)))
        owner = _EvidenceOwner(self._L, out)    # frees the C buffers when the arrays below are gone
        off = np.ctypeslib.as_array(out.rec_offset, shape=(n + 1,)).copy()
        total = int(off[-1])
        skipped = np.ctypeslib.as_array(out.skipped, shape=(max(n, 1),))[:n].copy()
        if total:
            raw = (C.c_uint8 * (total * RECORD_DTYPE.itemsize)).from_address(out.records)
            raw._svt_owner = owner
            recs = np.frombuffer(raw, dtype=RECORD_DTYPE)
        else:
            recs = np.zeros(0, RECORD_DTYPE)
        return off, recs, skipped

    def summarise(self, windows: np.ndarray, breakpoints: np.ndarray, read_groups: Sequence[str],
                  read_group_lib: Sequence[int], max_reads: Optional[int], count_mode: int,
                  n_threads: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(frag_offset uint64 [n+1], fragments FRAGMENT_DTYPE, skipped uint8 [n])"""
        windows = np.ascontiguousarray(windows, dtype=FETCH_DTYPE)
        breakpoints = np.ascontiguousarray(breakpoints, dtype=BREAKPOINT_DTYPE)
        n = int(windows.shape[0])
        if breakpoints.shape[0] != n:
            raise ValueError("windows and breakpoints must have the same length")
        names = (C.c_char_p * max(1, len(read_groups)))(*[rg.encode() for rg in read_groups])
        libs = (C.c_int32 * max(1, len(read_groups)))(*[int(x) for x in read_group_lib])
        a = _Args(n, windows.ctypes.data, breakpoints.ctypes.data, len(read_groups), names, libs,
                  -1 if max_reads is None else int(max_reads), int(count_mode), int(n_threads))
        out = _Summaries()
        hip._check(self._L.svt_bam_summarise(self._h, C.byref(a), C.byref(out)))
        owner = _SummariesOwner(self._L, out)   # frees the C buffers when the arrays below are gone
        off = np.ctypeslib.as_array(out.frag_offset, shape=(n + 1,)).copy()
        total = int(off[-1])
        skipped = np.ctypeslib.as_array(out.skipped, shape=(max(n, 1),))[:n].copy()
        if total:   # zero-copy view of the C array (1.4 GB for 10 M fragments)
            raw = (C.c_uint8 * (total * FRAGMENT_DTYPE.itemsize)).from_address(out.fragments)
            raw._svt_owner = owner   # every numpy view keeps `raw` alive through .base, and raw keeps the owner
            frags = np.frombuffer(raw, dtype=FRAGMENT_DTYPE)
        else:
            frags = np.zeros(0, FRAGMENT_DTYPE)
        return off, frags, skipped


class _EvidenceOwner:
    def __init__(self, lib, evidence):
        self._lib, self._e = lib, evidence

    def __del__(self):
        try:
            self._lib.svt_evidence_free(C.byref(self._e))
        except Exception:
            pass


class _SummariesOwner:
    def __init__(self, lib, summaries):
        self._lib, self._s = lib, summaries

    def __del__(self):
        try:
            self._lib.svt_summaries_free(C.byref(self._s))
        except Exception:
            pass
