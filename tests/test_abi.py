"""The C-ABI library loads and exports every symbol include/svtyper_hip.h declares (no compute:
runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="svtyper_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svt_[a-z_]+)\s*\(", src)))


def test_header_symbols_exported():
    from svtyper_amd import hip
    hip.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s
    assert sorted(hip.EXPORTS) == syms
    reads = declared_symbols("svtyper_reads.h")
    assert len(reads) >= 9
    for s in reads:
        assert hasattr(lib, s), "missing export %s" % s


def test_version_and_loud_failure_without_gpu():
    from svtyper_amd import hip, synth
    L = hip.load()
    assert L.svt_version() == hip.ABI_VERSION
    if hip.device_count() == 0:
        import pytest
        b = synth.make_units(10, 1, [synth.normal_library(n=20000)])
        with pytest.raises(hip.SvtyperHipError) as e:
            hip.genotype_batch(b)
        assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)


def test_native_sample_column_text_equals_python_formatting():
    """svt_format_results (host-only) == results.results_to_dicts + vcf.Genotype.get_gt_string, for every GT
    code, both skip conventions and several FORMAT orders (including keys the sample has no value for)."""
    import numpy as np
    from svtyper_amd import evidence as ev, hip, synth
    from svtyper_amd.results import results_to_dicts
    from oracle import c_oracle
    lib = synth.normal_library(n=30000)
    res = c_oracle.genotype_batch(synth.make_edge_cases([lib], seed=3), 0)
    codes = set(np.unique(res.gt).tolist())
    assert {ev.GT_BLANK, ev.GT_SKIPPED, ev.GT_MISSING, 0, 1, 2} <= codes
    ours = ("GT", "GQ", "SQ", "GL", "DP", "RO", "AO", "QR", "QA", "RS", "AS", "ASC", "RP", "AP", "AB")
    orders = [ours, ("GT", "SU", "GQ", "DP", "CN", "SQ", "GL", "RO", "AO", "QR", "QA", "RS", "AS", "ASC", "RP", "AP", "AB"),
              ("GT",), ("AB", "GL", "GT")]
    dicts = results_to_dicts(res)
    gts = res.gt.tolist()

    def cell(v):
        return "%0.2f" % v if type(v) == float else str(v)

    for fields in orders:
        for skipped_as_dots in (False, True):
            got = hip.format_results(res, fields, skipped_as_dots)
            assert len(got) == res.n_units
            for i, (d, gt) in enumerate(zip(dicts, gts)):
                if gt == ev.GT_SKIPPED and skipped_as_dots:       # classic.py:282-284: only GT is set
                    want = ":".join("./." if f == "GT" else "." for f in fields)
                else:
                    want = ":".join(cell(d["formats"][f]) if f in d["formats"] else "." for f in fields)
                assert got[i] == want, (i, gt, fields, got[i], want)
