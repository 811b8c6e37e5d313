"""ctypes front-end of oracle/svt_oracle.c (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from svtyper_amd.evidence import (CEvidenceBatch, CLibrary, EvidenceBatch, LibraryTable,
                                  N_COUNTS, N_TALLIES, Results)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsvt_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile svt_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "svt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsvt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.svt_oracle_prob_mapq.restype = C.c_double
        L.svt_oracle_prob_mapq.argtypes = [C.c_int]
        L.svt_oracle_log_choose.restype = C.c_double
        L.svt_oracle_log_choose.argtypes = [C.c_int64, C.c_int64]
        L.svt_oracle_bayes_gt.restype = None
        L.svt_oracle_bayes_gt.argtypes = [C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_double)]
        L.svt_oracle_p_concordant.restype = C.c_int
        L.svt_oracle_p_concordant.argtypes = [C.POINTER(CLibrary), C.c_uint64, C.c_int32, C.c_int,
                                              C.c_int32]
        L.svt_oracle_genotype.restype = None
        L.svt_oracle_genotype.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int8)]
        L.svt_oracle_batch.restype = C.c_int
        L.svt_oracle_batch.argtypes = [C.POINTER(CEvidenceBatch), C.c_void_p, C.c_uint, C.c_int]
        L.svt_oracle_threads.restype = C.c_int
        _lib = L
    return _lib


def prob_mapq(q: int) -> float:
    return lib().svt_oracle_prob_mapq(int(q))


def log_choose(n: int, k: int) -> float:
    return lib().svt_oracle_log_choose(int(n), int(k))


def bayes_gt(ref: int, alt: int, is_dup: bool):
    out = (C.c_double * 3)()
    lib().svt_oracle_bayes_gt(int(ref), int(alt), int(bool(is_dup)), out)
    return (out[0], out[1], out[2])


def p_concordant(table: LibraryTable, ospan_length: int, var_length=None) -> bool:
    h = np.ascontiguousarray(table.hist, dtype=np.uint32)
    cl = CLibrary(h.ctypes.data_as(C.POINTER(C.c_uint32)), int(table.key_min), int(h.shape[0]),
                  float(table.mean), float(table.sd))
    return bool(lib().svt_oracle_p_concordant(C.byref(cl), int(h.sum(dtype=np.uint64)),
                                              int(ospan_length), int(var_length is not None),
                                              int(var_length or 0)))


def genotype_from_tallies(tallies, svtype: int, split_weight=1.0, disc_weight=1.0):
    """tallies in TALLY_NAMES order (ref_seq, alt_seq, alt_clip, ref_span, alt_span)."""
    t = (C.c_double * N_TALLIES)(*[float(x) for x in tallies])
    gl = (C.c_double * 3)()
    sq = C.c_double()
    counts = (C.c_int32 * N_COUNTS)()
    gt = C.c_int8()
    lib().svt_oracle_genotype(t, int(svtype), float(split_weight), float(disc_weight), gl,
                              C.byref(sq), counts, C.byref(gt))
    return dict(gl=(gl[0], gl[1], gl[2]), sq=sq.value, counts=list(counts), gt=gt.value)


def genotype_batch(batch: EvidenceBatch, flags: int = 0, n_threads: int = 0, out: Results = None) -> Results:
    """`out` may be a preallocated Results (bench.py times only the C call with it)."""
    if out is None:
        out = Results.empty(batch.n_units)
    cb = batch.as_c()
    rc = lib().svt_oracle_batch(C.byref(cb), C.c_void_p(out.ptr()), int(flags), int(n_threads))
    if rc != 0:
        raise RuntimeError("svt_oracle_batch failed: %d" % rc)
    return out


def max_threads() -> int:
    return lib().svt_oracle_threads()
