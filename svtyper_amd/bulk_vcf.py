"""ctypes binding of the bulk VCF parse / emit calls (include/svtyper_vcf.h).

A block of variant lines becomes breakpoint ARRAYS in one native call, and once the device has the
results the output lines of the block are one text -- no `Variant`, no breakpoint dict and no join per
line (what svtyper/parsers.py:256-399, classic.py:219-278 and singlesample.py:577-652 do per line).
`vcf.Variant` objects are still made for the lines the parser hands back (line kind PYTHON) and for
everything once it stops in front of a BND line it cannot express (`consumed` < len): the drivers'
per-line code is the general implementation and the checker of this one (tests/test_bulk_vcf.py).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

from . import hip

LINE_SITE, LINE_HELD, LINE_PYTHON, LINE_SKIPPED = 0, 1, 2, 3
SUM_QUALS, SKIP_HASH_LINES = 1, 2
QUAL_SSO, QUAL_CLASSIC = 0, 1
TEXT_ENCODING = ("utf-8", "surrogateescape")      # whatever bytes the text layer let through travel back unchanged


class _View(C.Structure):
    _fields_ = [("n_lines", C.c_uint64), ("line_kind", C.c_void_p), ("line_begin", C.c_void_p), ("line_site", C.c_void_p),
                ("n_sites", C.c_uint64), ("chrom_a", C.c_void_p), ("chrom_b", C.c_void_p), ("pos_a", C.c_void_p),
                ("pos_b", C.c_void_p), ("ci", C.c_void_p), ("var_length", C.c_void_p), ("svtype", C.c_void_p),
                ("strands", C.c_void_p), ("qual_in", C.c_void_p)]


EXPORTS = ("svt_vcf_parser_create", "svt_vcf_parser_free", "svt_vcf_parser_n_chroms", "svt_vcf_parser_chrom",
           "svt_vcf_parser_n_pending", "svt_vcf_parser_pending_line", "svt_vcf_parse", "svt_vcf_chunk_free",
           "svt_vcf_chunk_view", "svt_vcf_emit")

_declared = False


def _lib():
    global _declared
    L = hip.load()
    if not _declared:
        L.svt_vcf_parser_create.restype = C.c_int
        L.svt_vcf_parser_create.argtypes = [C.POINTER(C.c_char_p), C.c_void_p, C.c_uint32, C.c_double, C.c_uint32,
                                            C.POINTER(C.c_void_p)]
        L.svt_vcf_parser_free.restype = None
        L.svt_vcf_parser_free.argtypes = [C.c_void_p]
        L.svt_vcf_parser_n_chroms.restype = C.c_uint32
        L.svt_vcf_parser_n_chroms.argtypes = [C.c_void_p]
        L.svt_vcf_parser_chrom.restype = C.c_char_p
        L.svt_vcf_parser_chrom.argtypes = [C.c_void_p, C.c_uint32]
        L.svt_vcf_parser_n_pending.restype = C.c_uint32
        L.svt_vcf_parser_n_pending.argtypes = [C.c_void_p]
        L.svt_vcf_parser_pending_line.restype = C.c_char_p
        L.svt_vcf_parser_pending_line.argtypes = [C.c_void_p, C.c_uint32]
        L.svt_vcf_parse.restype = C.c_int
        L.svt_vcf_parse.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.svt_vcf_chunk_free.restype = None
        L.svt_vcf_chunk_free.argtypes = [C.c_void_p]
        L.svt_vcf_chunk_view.restype = C.c_int
        L.svt_vcf_chunk_view.argtypes = [C.c_void_p, C.POINTER(_View)]
        L.svt_vcf_emit.restype = C.c_int
        L.svt_vcf_emit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_char_p,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        _declared = True
    return L


def available() -> bool:
    """the library is built and exports the bulk VCF calls (no GPU needed for them)"""
    try:
        return hasattr(_lib(), "svt_vcf_parse")
    except (hip.SvtyperHipError, OSError):
        return False


class SiteArrays:
    """Breakpoints of n sites as arrays (what Vcf.get_variant_breakpoints returns per site as a dict): `names` is a
    table of chromosome names, `chrom` [n, 2] indexes it (side A, side B); pos [n, 2], ci [n, 4] = A lo, A hi, B lo,
    B hi; reverse [n] bit 0 = A, bit 1 = B; svtype [n] (evidence.SVTYPE_CODE); var_length [n] (DEL, else 0)."""

    __slots__ = ("names", "chrom", "pos", "ci", "reverse", "svtype", "var_length")

    def __init__(self, names, chrom, pos, ci, reverse, svtype, var_length):
        self.names, self.chrom, self.pos, self.ci = list(names), chrom, pos, ci
        self.reverse, self.svtype, self.var_length = reverse, svtype, var_length

    def __len__(self):
        return int(self.pos.shape[0])

    @classmethod
    def empty(cls) -> "SiteArrays":
        return cls([], np.zeros((0, 2), np.int32), np.zeros((0, 2), np.int64), np.zeros((0, 4), np.int64),
                   np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.int64))

    @classmethod
    def from_dicts(cls, sites: List[dict]) -> "SiteArrays":
        """the breakpoint dicts of the per-line code as arrays (C-level iteration: itemgetter + fromiter)"""
        from itertools import chain
        from operator import itemgetter, methodcaller
        from . import evidence as ev
        n = len(sites)
        if n == 0:
            return cls.empty()
        get = lambda key, seq: map(itemgetter(key), seq)
        A = list(get("A", sites))
        B = list(get("B", sites))
        pos = np.empty((n, 2), np.int64)
        pos[:, 0] = np.fromiter(get("pos", A), np.int64, n)
        pos[:, 1] = np.fromiter(get("pos", B), np.int64, n)
        ci = np.empty((n, 4), np.int64)
        for col, side in ((0, A), (2, B)):      # (a ci that is not a pair fails here)
            ci[:, col:col + 2] = np.fromiter(chain.from_iterable(get("ci", side)), np.int64, 2 * n).reshape(n, 2)
        rev = np.fromiter(get("is_reverse", A), np.bool_, n).astype(np.uint8)
        rev |= np.fromiter(get("is_reverse", B), np.bool_, n).astype(np.uint8) << 1
        svt = np.fromiter(map(ev.SVTYPE_CODE.__getitem__, get("svtype", sites)), np.uint8, n)
        vlen = np.fromiter(map(methodcaller("get", "var_length", 0), sites), np.int64, n)
        vlen[svt != ev.SVTYPE_CODE["DEL"]] = 0
        chrom_a, chrom_b = list(get("chrom", A)), list(get("chrom", B))
        names = sorted(set(chrom_a).union(chrom_b))
        index = {c: i for i, c in enumerate(names)}.__getitem__
        chrom = np.empty((n, 2), np.int32)
        chrom[:, 0] = np.fromiter(map(index, chrom_a), np.int32, n)
        chrom[:, 1] = np.fromiter(map(index, chrom_b), np.int32, n)
        return cls(names, chrom, pos, ci, rev, svt, vlen)

    @classmethod
    def concat(cls, parts: List["SiteArrays"]) -> "SiteArrays":
        parts = [p for p in parts if len(p)]
        if not parts:
            return cls.empty()
        if len(parts) == 1:
            return parts[0]
        names, chroms = [], []
        for p in parts:
            chroms.append(p.chrom + np.int32(len(names)))
            names.extend(p.names)
        cat = lambda f: np.concatenate([getattr(p, f) for p in parts])
        return cls(names, np.concatenate(chroms), cat("pos"), cat("ci"), cat("reverse"), cat("svtype"), cat("var_length"))


class VcfChunk:
    """The parsed lines of one block (svt_vcf_chunk): line kinds, the breakpoints of its sites, and -- emit() -- the
    output lines of those sites from their result records."""

    def __init__(self, lib, handle, parser: "VcfParser"):
        self._L, self._h = lib, handle
        v = _View()
        hip._check(lib.svt_vcf_chunk_view(handle, C.byref(v)))
        n, s = int(v.n_lines), int(v.n_sites)
        arr = lambda ptr, ct, shape: (np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=shape).copy() if shape[0] else
                                      np.zeros(shape, np.dtype(ct)))
        self.n_lines, self.n_sites = n, s
        self.line_kind = arr(v.line_kind, C.c_uint8, (n,))
        self.line_begin = arr(v.line_begin, C.c_uint64, (n + 1,)).astype(np.int64)
        self.line_site = arr(v.line_site, C.c_uint32, (n,))
        chrom = np.empty((s, 2), np.int32)
        chrom[:, 0] = arr(v.chrom_a, C.c_int32, (s,))
        chrom[:, 1] = arr(v.chrom_b, C.c_int32, (s,))
        pos = np.empty((s, 2), np.int64)
        pos[:, 0] = arr(v.pos_a, C.c_int64, (s,))
        pos[:, 1] = arr(v.pos_b, C.c_int64, (s,))
        self.qual_in = arr(v.qual_in, C.c_double, (s,))
        self.sites = SiteArrays(parser.chrom_names(), chrom, pos, arr(v.ci, C.c_int64, (s * 4,)).reshape(s, 4),
                                arr(v.strands, C.c_uint8, (s,)), arr(v.svtype, C.c_uint8, (s,)), arr(v.var_length, C.c_int64, (s,)))

    def emit(self, results, n_samples: int, qual_mode: int, fields, skipped_as_dots: bool, format_string: str) -> Tuple[bytes, np.ndarray]:
        """(text, site_offset[n_sites + 1]): the output lines of every site; `results`: the sites' units, site-major,
        SQ refined (hip.host_sq)."""
        if results.n_units < self.n_sites * n_samples:
            raise ValueError("%d result records for %d sites x %d samples" % (results.n_units, self.n_sites, n_samples))
        codes = np.array([hip.FORMAT_CODES.get(f, hip.FORMAT_ABSENT) for f in fields], dtype=np.uint8)
        text, off = C.c_void_p(), C.c_void_p()
        hip._check(self._L.svt_vcf_emit(self._h, C.c_void_p(results.ptr()), int(n_samples), int(qual_mode), codes.ctypes.data,
                                        len(codes), 1 if skipped_as_dots else 0, format_string.encode("ascii"),
                                        C.byref(text), C.byref(off)))
        try:
            o = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), shape=(self.n_sites + 1,)).astype(np.int64)
            return C.string_at(text, int(o[-1])), o
        finally:
            self._L.svt_format_free(text, off)

    def close(self):
        if self._h:
            self._L.svt_vcf_chunk_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VcfParser:
    """svt_vcf_parser: the header's INFO declarations (order, Flag types), --max_ci_dist, the chromosome table and the
    BND mates still waiting for their partner."""

    def __init__(self, vcf, max_ci_dist, sum_quals: bool, skip_hash_lines: bool):
        L = _lib()
        self._L = L
        ids = [h.id.encode(*TEXT_ENCODING) for h in vcf.info_list]
        flags = np.array([1 if h.type == "Flag" else 0 for h in vcf.info_list], dtype=np.uint8)
        arr = (C.c_char_p * max(1, len(ids)))(*ids)
        self._h = C.c_void_p()
        hip._check(L.svt_vcf_parser_create(arr, flags.ctypes.data, len(ids), float(max_ci_dist),
                                           (SUM_QUALS if sum_quals else 0) | (SKIP_HASH_LINES if skip_hash_lines else 0),
                                           C.byref(self._h)))
        self._names: List[str] = []

    def chrom_names(self) -> List[str]:
        n = int(self._L.svt_vcf_parser_n_chroms(self._h))
        while len(self._names) < n:
            self._names.append(self._L.svt_vcf_parser_chrom(self._h, len(self._names)).decode(*TEXT_ENCODING))
        return self._names

    def pending_lines(self) -> List[str]:
        n = int(self._L.svt_vcf_parser_n_pending(self._h))
        return [self._L.svt_vcf_parser_pending_line(self._h, i).decode(*TEXT_ENCODING) for i in range(n)]

    def parse(self, data: bytes) -> Tuple[VcfChunk, int]:
        """(chunk, bytes consumed): consumed < len(data) when the parse stopped in front of a BND line only the per-line
        code can handle -- the caller goes on with that code from there (seeded with pending_lines())."""
        h, used = C.c_void_p(), C.c_size_t()
        hip._check(self._L.svt_vcf_parse(self._h, data, len(data), C.byref(h), C.byref(used)))
        return VcfChunk(self._L, h, self), int(used.value)

    def close(self):
        if self._h:
            self._L.svt_vcf_parser_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
