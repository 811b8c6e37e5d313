"""Device result records -> the reference's per-sample result (FORMAT values).

Pure formatting of what the kernels computed: '%.0f' GL strings, '%.2g' allele balance, the
'.'-for-missing conventions of svtyper/classic.py:454-513 and svtyper/singlesample.py:207-227,
430-471.  No likelihood arithmetic happens here.
"""
from __future__ import annotations

from . import evidence as ev

FORMAT_ORDER = ("GT", "GQ", "SQ", "GL", "DP", "AO", "RO", "AS", "ASC", "RS", "AP", "RP", "QR", "QA", "AB")
_CNT = {name: i for i, name in enumerate(ev.COUNT_NAMES)}


def blank_result() -> dict:
    """singlesample.py:207-227 / classic.py:496-513"""
    return {"qual": 0, "formats": {"GT": "./.", "GQ": ".", "SQ": ".", "GL": ".", "DP": 0, "AO": 0, "RO": 0,
                                   "AS": 0, "ASC": 0, "RS": 0, "AP": 0, "RP": 0, "QR": 0, "QA": 0, "AB": "."}}


def result_from_record(rec) -> dict:
    """One element of Results.rec -> {'qual': float|0, 'formats': {...}} shaped like the output of
    svtyper/singlesample.py:406-473 bayesian_genotype()."""
    gt = int(rec["gt"])
    if gt in (ev.GT_BLANK, ev.GT_SKIPPED):
        return blank_result()
    c = rec["counts"]
    out = blank_result()
    f = out["formats"]
    f["GL"] = ",".join("%.0f" % float(x) for x in rec["gl"])           # classic.py:454
    for name in ("DP", "RO", "AO", "QR", "QA", "RS", "AS", "ASC", "RP", "AP"):
        f[name] = int(c[_CNT[name]])
    qr, qa = f["QR"], f["QA"]
    f["AB"] = "%.2g" % (qa / float(qr + qa)) if (qr + qa) != 0 else "."  # classic.py:466-469
    if gt >= 0:
        sq = float(rec["sq"])
        f["GQ"] = int(c[_CNT["GQ"]])
        f["SQ"] = sq
        f["GT"] = ev.GT_STRING[gt]
        out["qual"] = sq
    return out


def results_to_dicts(results) -> list:
    """All records of a Results at once (same dicts as result_from_record, one bulk conversion to
    Python objects instead of one numpy scalar access per field)."""
    rec = results.rec
    gts = rec["gt"].tolist()
    gls = rec["gl"].tolist()
    sqs = rec["sq"].tolist()
    cnts = rec["counts"].tolist()
    out = []
    for gt, gl, sq, c in zip(gts, gls, sqs, cnts):
        res = blank_result()
        if gt not in (ev.GT_BLANK, ev.GT_SKIPPED):
            f = res["formats"]
            f["GL"] = "%.0f,%.0f,%.0f" % (gl[0], gl[1], gl[2])
            (f["QR"], f["QA"], gq, f["DP"], f["RO"], f["AO"], f["RS"], f["AS"], f["ASC"], f["RP"], f["AP"]) = c
            tot = f["QR"] + f["QA"]
            f["AB"] = "%.2g" % (f["QA"] / float(tot)) if tot != 0 else "."
            if gt >= 0:
                f["GQ"], f["SQ"], f["GT"] = gq, sq, ev.GT_STRING[gt]
                res["qual"] = sq
        out.append(res)
    return out
