"""Pure-Python restatement of the SVTyper likelihood hot path -- TEST INFRASTRUCTURE ONLY.

Statement-for-statement after the reference (CPython floats, `math.log(x, 10)`, `10 ** x`), over
the packed evidence records of include/svtyper_hip.h.  It is the second, independent restatement
next to svt_oracle.c (the two are cross-checked in tests/test_oracle_golden.py) and the closest
stand-in for "the reference's own CPU path" that can run on the GPU box, where the reference itself
is not available: bench.py can time it (1 process and a multiprocessing.Pool, mirroring
svtyper/singlesample.py:746-748) next to the C port.

Every function cites the reference lines it follows (paths relative to the reference checkout).
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np

from svtyper_amd import evidence as ev
from svtyper_amd.evidence import EvidenceBatch, Results


def prob_mapq(mapq: int) -> float:
    """svtyper/utils.py:74-75"""
    return 1 - 10 ** (-mapq / 10.0)


def log_choose(n: int, k: int) -> float:
    """svtyper/statistics.py:9-20"""
    r = 0.0
    if k * 2 > n:
        k = n - k
    for d in range(1, k + 1):
        r += math.log(n, 10)
        r -= math.log(d, 10)
        n -= 1
    return r


def bayes_gt(ref: int, alt: int, is_dup: bool):
    """svtyper/statistics.py:23-37"""
    if is_dup:
        p_alt = [1e-2, 0.2, 1 / 3.0]
    else:
        p_alt = [1e-3, 0.5, 0.9]
    total = ref + alt
    log_combo = log_choose(total, alt)
    lp_homref = log_combo + alt * math.log(p_alt[0], 10) + ref * math.log(1 - p_alt[0], 10)
    lp_het = log_combo + alt * math.log(p_alt[1], 10) + ref * math.log(1 - p_alt[1], 10)
    lp_homalt = log_combo + alt * math.log(p_alt[2], 10) + ref * math.log(1 - p_alt[2], 10)
    return (lp_homref, lp_het, lp_homalt)


class _Lib:
    """Library view with the Counter semantics of svtyper/parsers.py:579-583 (missing key -> 0)."""

    def __init__(self, table: ev.LibraryTable):
        self.mean = table.mean
        self.sd = table.sd
        n = int(table.hist.sum(dtype=np.uint64))
        self.dens = {int(table.key_min) + i: float(h) / n for i, h in enumerate(table.hist.tolist()) if h}

    def density(self, key) -> float:
        return self.dens.get(key, 0)


def p_concordant(lib: _Lib, ospan_length: int, var_length=None) -> bool:
    """svtyper/parsers.py:861-882 (Python 2: `None > 0.5` is False)"""
    disc_prior = 0.05
    conc_prior = 1 - disc_prior
    z = 3
    if var_length is None:
        var_length = lib.mean + lib.sd * z
    try:
        p = float(lib.density(ospan_length)) * conc_prior / (
            conc_prior * lib.density(ospan_length) + disc_prior * (lib.density(ospan_length - var_length)))
    except ZeroDivisionError:
        return False
    return p > 0.5


def tally_unit(unit, recs: Sequence, libs: List[_Lib], sso: bool):
    """svtyper/classic.py:286-435 (sso: svtyper/singlesample.py:246-404).  Returns the five tallies
    after the zeroing rules, in ev.TALLY_NAMES order."""
    ref_span, alt_span = 0, 0
    ref_seq, alt_seq = 0, 0
    alt_clip = 0
    l_ref_seq = l_alt_seq = l_alt_clip = 0
    is_del = unit["svtype"] == 0
    var_length = int(unit["var_length"]) if is_del else None
    pos_delta = int(unit["pos_delta"])
    first = True
    for (ospan, mq_a, mq_b, rs_a, rs_b, seq_l, seq_r, clip_l, clip_r, flags) in recs:
        lib = libs[(flags >> ev.REC_LIB_SHIFT) & 0xFFFF]
        if sso and not (flags & ev.REC_CONTINUATION):
            if not first:                                            # singlesample.py:370-372
                ref_seq += l_ref_seq
                alt_seq += l_alt_seq
                alt_clip += l_alt_clip
            l_ref_seq = l_alt_seq = l_alt_clip = 0
        first = False
        # classic.py:306-311 (gated MAPQ: prob_mapq(0) == 0.0)
        if sso:
            l_ref_seq += prob_mapq(rs_a)
            l_ref_seq += prob_mapq(rs_b)
        else:
            ref_seq += prob_mapq(rs_a)
            ref_seq += prob_mapq(rs_b)
        # classic.py:317-328
        p_seq = (prob_mapq(seq_l) + prob_mapq(seq_r)) / 2.0
        p_clip = (prob_mapq(clip_l) + prob_mapq(clip_r)) / 2.0
        if sso:
            l_alt_seq += p_seq
            l_alt_clip += p_clip
        else:
            alt_seq += p_seq
            alt_clip += p_clip
        # classic.py:339-408
        small_del = is_del and pos_delta < 2 * lib.sd
        alt_straddle = (not small_del) and bool(flags & ev.REC_ALT_STRADDLE)
        if alt_straddle:
            if is_del:
                p_conc = p_concordant(lib, ospan, var_length)
                alt_span += (1 - p_conc) * prob_mapq(mq_a) * prob_mapq(mq_b)
            else:
                alt_span += prob_mapq(mq_a) * prob_mapq(mq_b)
        ref_straddle_A = (not small_del) and bool(flags & ev.REC_REF_STRADDLE_A)
        ref_straddle_B = (not small_del) and bool(flags & ev.REC_REF_STRADDLE_B)
        if ref_straddle_A or ref_straddle_B:
            if not (ref_straddle_A and ref_straddle_B) or is_del:
                p_conc = p_concordant(lib, ospan, var_length)
                p_reference = p_conc * prob_mapq(mq_a) * prob_mapq(mq_b)
                ref_span += (ref_straddle_A + ref_straddle_B) * p_reference / 2
    if sso and not first:
        ref_seq += l_ref_seq
        alt_seq += l_alt_seq
        alt_clip += l_alt_clip
    # classic.py:425-435
    if (alt_seq + alt_clip) < 0.5 and alt_span >= 1:
        alt_seq = 0
        alt_clip = 0
        ref_seq = 0
    if alt_span < 0.5 and (alt_seq + alt_clip) >= 1:
        alt_span = 0
        ref_span = 0
    if alt_span + alt_seq == 0 and alt_clip > 0:
        alt_clip = 0
    return float(ref_seq), float(alt_seq), float(alt_clip), float(ref_span), float(alt_span)


def genotype_tallies(t, svtype: int, split_weight, disc_weight, out) -> None:
    """svtyper/classic.py:437-513 into one RESULT_DTYPE element."""
    ref_seq, alt_seq, alt_clip, ref_span, alt_span = t
    out["tallies"] = t
    out["counts"][ev.COUNT_NAMES.index("GQ")] = -1
    if not (ref_seq + alt_seq + ref_span + alt_span + alt_clip > 0):
        out["gt"] = ev.GT_BLANK
        return
    is_dup = svtype == 1
    alt_splitters = alt_seq + alt_clip
    QR = int(split_weight * ref_seq) + int(disc_weight * ref_span)
    QA = int(split_weight * alt_splitters) + int(disc_weight * alt_span)
    gt_lplist = bayes_gt(QR, QA, is_dup)
    best, second_best = sorted([(i, e) for i, e in enumerate(gt_lplist)], key=lambda x: x[1], reverse=True)[0:2]
    c = out["counts"]
    for name, val in (("QR", QR), ("QA", QA), ("DP", int(ref_seq + alt_seq + alt_clip + ref_span + alt_span)),
                      ("RO", int(ref_seq + ref_span)), ("AO", int(alt_seq + alt_clip + alt_span)),
                      ("RS", int(ref_seq)), ("AS", int(alt_seq)), ("ASC", int(alt_clip)), ("RP", int(ref_span)),
                      ("AP", int(alt_span))):
        c[ev.COUNT_NAMES.index(name)] = val
    out["gl"] = gt_lplist
    gt_sum = 0
    for gt in gt_lplist:
        try:
            gt_sum += 10 ** gt
        except OverflowError:
            gt_sum += 0
    if gt_sum > 0:
        gt_sum_log = math.log(gt_sum, 10)
        out["sq"] = abs(-10 * (gt_lplist[0] - gt_sum_log))
        c[ev.COUNT_NAMES.index("GQ")] = int(min(-10 * (second_best[1] - best[1]), 200))
        out["gt"] = best[0]
    else:
        out["gt"] = ev.GT_MISSING


def genotype_batch(batch: EvidenceBatch, flags: int = 0, lo: int = 0, hi: int = None) -> Results:
    """Units [lo, hi) of `batch` (all by default)."""
    hi = batch.n_units if hi is None else hi
    sso = bool(flags & ev.FLAG_SSO_ASSOCIATION)
    libs = [_Lib(t) for t in batch.libs]
    res = Results.empty(hi - lo)
    offs = batch.rec_offset
    for k, u in enumerate(range(lo, hi)):
        unit = batch.units[u]
        out = res.rec[k]
        if unit["flags"] & ev.UNIT_SKIP:
            out["gt"] = ev.GT_SKIPPED
            out["counts"][ev.COUNT_NAMES.index("GQ")] = -1
            continue
        recs = batch.records[int(offs[u]):int(offs[u + 1])].tolist()
        t = tally_unit(unit, recs, libs, sso)
        genotype_tallies(t, int(unit["svtype"]), batch.split_weight, batch.disc_weight, out)
    return res


def _pool_task(args):
    batch, flags, lo, hi = args
    return genotype_batch(batch, flags, lo, hi).rec


def genotype_batch_pool(batch: EvidenceBatch, flags: int = 0, processes: int = 2, batch_size: int = 1000) -> Results:
    """multiprocessing.Pool over batches of `batch_size` units, results in submission order
    (the parallel structure of svtyper/singlesample.py:723-751)."""
    import multiprocessing as mp
    tasks = []
    for lo in range(0, batch.n_units, batch_size):
        hi = min(batch.n_units, lo + batch_size)
        tasks.append((batch.slice(lo, hi), flags, 0, hi - lo))
    with mp.get_context("fork").Pool(processes) as pool:
        parts = pool.map(_pool_task, tasks)
    return Results(np.concatenate(parts) if parts else np.zeros(0, ev.RESULT_DTYPE))
