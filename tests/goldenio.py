"""Readers for the committed golden vectors (tests/golden/*.json.gz) -- TEST INFRASTRUCTURE."""
import gzip
import json
import os

import numpy as np

from svtyper_amd import evidence as ev
from svtyper_amd.evidence import EvidenceBatch, LibraryTable, RECORD_DTYPE, UNIT_DTYPE

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TALLIES = ev.TALLY_NAMES


def load(name):
    with gzip.open(os.path.join(GOLDEN, name), "rb") as f:
        return json.loads(f.read().decode())


def fh(x):
    return float.fromhex(x)


def libraries(libs_json):
    return [LibraryTable.from_counter({int(k): int(v) for k, v in L["hist"].items()}, fh(L["mean"]), fh(L["sd"]),
                                      L["name"]) for L in libs_json]


def unit_from_breakpoint(bp):
    u = np.zeros(1, UNIT_DTYPE)
    u["svtype"] = ev.SVTYPE_CODE[bp["svtype"]]
    if bp["svtype"] == "DEL":
        u["var_length"] = bp["var_length"]
    u["pos_delta"] = max(-2**31, min(2**31 - 1, bp["B"]["pos"] - bp["A"]["pos"]))
    return u


def records_from_rows(rows):
    rec = np.zeros(len(rows), RECORD_DTYPE)
    if rows:
        arr = np.asarray(rows, dtype=np.int64)
        for i, name in enumerate(RECORD_DTYPE.names):
            rec[name] = arr[:, i]
    return rec


def batch_from_sites(sites, libs_json, split_weight=1.0, disc_weight=1.0):
    offs = [0]
    units, recs = [], []
    for s in sites:
        r = records_from_rows(s["records"])
        units.append(unit_from_breakpoint(s["breakpoint"]))
        recs.append(r)
        offs.append(offs[-1] + len(r))
    return EvidenceBatch(np.asarray(offs, np.uint64), np.concatenate(units), np.concatenate(recs),
                         libraries(libs_json), split_weight, disc_weight)


def golden_result(res_json):
    """json -> the reference's result dict (floats restored)."""
    out = {"qual": fh(res_json["qual"]) if isinstance(res_json["qual"], str) else res_json["qual"], "formats": {}}
    for k, v in res_json["formats"].items():
        out["formats"][k] = fh(v["f"]) if isinstance(v, dict) else v
    return out


def apply_zeroing(t):
    """svtyper/classic.py:425-435 on a dict of the five raw tallies (test helper)."""
    t = dict(t)
    if (t["alt_seq"] + t["alt_clip"]) < 0.5 and t["alt_span"] >= 1:
        t["alt_seq"] = 0; t["alt_clip"] = 0; t["ref_seq"] = 0
    if t["alt_span"] < 0.5 and (t["alt_seq"] + t["alt_clip"]) >= 1:
        t["alt_span"] = 0; t["ref_span"] = 0
    if t["alt_span"] + t["alt_seq"] == 0 and t["alt_clip"] > 0:
        t["alt_clip"] = 0
    return t


def assert_result_equal(got: dict, want: dict, sq_tol=0.0, where=""):
    """Exact equality of every FORMAT value; SQ / qual within sq_tol (0 = bit-exact)."""
    gf, wf = got["formats"], want["formats"]
    assert set(gf) == set(wf), where
    for k in wf:
        if k == "SQ" and isinstance(wf[k], float):
            assert isinstance(gf[k], float) and abs(gf[k] - wf[k]) <= sq_tol, (where, k, gf[k], wf[k])
        else:
            assert gf[k] == wf[k] and type(gf[k]) == type(wf[k]), (where, k, gf[k], wf[k])
    if isinstance(want["qual"], float):
        assert abs(got["qual"] - want["qual"]) <= sq_tol, (where, got["qual"], want["qual"])
    else:
        assert got["qual"] == want["qual"], where
