"""The C-ABI library loads and exports every symbol include/svtyper_hip.h declares (no compute:
runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="svtyper_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svt_[a-z_0-9]+)\s*\(", src)))


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


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every struct of include/svtyper_hip.h, as a C compiler lays them out, against the numpy
    dtypes and ctypes Structures the Python side hands to the library (and the INTEGRATION.md stub copies)."""
    import subprocess
    import numpy as np
    from svtyper_amd import evidence as ev, geometry as geo
    structs = {
        "svt_record": ["ospan_len", "mapq_a", "mapq_b", "rs_a", "rs_b", "seq_l", "seq_r", "clip_l", "clip_r", "flags"],
        "svt_unit": ["var_length", "pos_delta", "sample", "svtype", "flags", "libs"],
        "svt_library": ["hist", "key_min", "n_bins", "mean", "sd"],
        "svt_evidence_batch": ["n_units", "rec_offset", "units", "records", "n_libs", "libs", "split_weight", "disc_weight"],
        "svt_read_summary": ["tid", "start", "end", "iv_start", "iv_end", "mapq", "flags", "reserved"],
        "svt_piece_summary": ["tid", "start", "end", "mapq", "flags", "reserved"],
        "svt_fragment": ["read", "seq", "clip"],
        "svt_breakpoint": ["tid_a", "pos_a", "ci_a", "tid_b", "pos_b", "ci_b", "var_length", "sample", "svtype", "flags", "reserved"],
        "svt_fragment_batch": ["n_units", "frag_offset", "breakpoints", "fragments", "n_libs", "libs", "split_weight",
                               "disc_weight", "min_aligned", "split_slop"],
        "svt_result": ["gl", "sq", "tallies", "counts", "gt", "pad"],
        "svt_result96": ["gl", "sq", "tallies", "qr", "qa", "gq", "gt", "pad", "unit", "pad2"],
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "svtyper_hip.h"', 'int main(void) {']
    for name, fields in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for f in fields:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    for macro in ("SVT_ABI_VERSION", "SVT_FLAG_SSO_ASSOCIATION", "SVT_FLAG_GENERAL_TABLES", "SVT_FLAG_RESULT96", "SVT_NO_UNIT", "SVT_REC_LIB_SHIFT", "SVT_REC_CONTINUATION", "SVT_REC_HAS_PAIR"):
        lines.append('printf("%s %%d\\n", (int)%s);' % (macro, macro))
    lines += ['return 0; }']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "probe")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    c = dict(l.split() for l in subprocess.check_output([exe], text=True).splitlines())
    c = {k: int(v) for k, v in c.items()}

    from svtyper_amd import hip
    assert c["SVT_ABI_VERSION"] == hip.ABI_VERSION
    assert (c["SVT_FLAG_SSO_ASSOCIATION"], c["SVT_FLAG_GENERAL_TABLES"]) == (ev.FLAG_SSO_ASSOCIATION, ev.FLAG_GENERAL_TABLES)
    assert (c["SVT_REC_LIB_SHIFT"], c["SVT_REC_CONTINUATION"], c["SVT_REC_HAS_PAIR"]) == (ev.REC_LIB_SHIFT, ev.REC_CONTINUATION, ev.REC_HAS_PAIR)
    assert c["SVT_FLAG_RESULT96"] == ev.FLAG_RESULT96 and c["svt_result96"] == 96
    # the 96-byte record is svt_result's first 84 bytes + gt: what svt_results_expand96 and the kernel's store rely on
    assert c["svt_result96.qr"] == c["svt_result.counts"] == 72 and c["svt_result96.gt"] == 84 and c["svt_result.gt"] == 116
    assert c["svt_result96.unit"] == 88 and c["SVT_NO_UNIT"] == ev.NO_UNIT - (1 << 32)
    dtypes = {"svt_record": ev.RECORD_DTYPE, "svt_unit": ev.UNIT_DTYPE, "svt_result": ev.RESULT_DTYPE, "svt_result96": ev.RESULT96_DTYPE,
              "svt_read_summary": geo.READ_DTYPE, "svt_piece_summary": geo.PIECE_DTYPE, "svt_fragment": geo.FRAGMENT_DTYPE,
              "svt_breakpoint": geo.BREAKPOINT_DTYPE}
    for name, dt in dtypes.items():
        assert dt.itemsize == c[name], (name, dt.itemsize, c[name])
        assert list(dt.names) == structs[name], (name, dt.names)
        for f in dt.names:
            assert dt.fields[f][1] == c["%s.%s" % (name, f)], (name, f, dt.fields[f][1], c["%s.%s" % (name, f)])
    assert c["svt_record"] == 16 and c["svt_unit"] == 16 and c["svt_result"] == 128 and c["svt_fragment"] == 128
    ctys = {"svt_library": ev.CLibrary, "svt_evidence_batch": ev.CEvidenceBatch, "svt_fragment_batch": geo.CFragmentBatch}
    for name, ct in ctys.items():
        assert ctypes.sizeof(ct) == c[name], (name, ctypes.sizeof(ct), c[name])
        assert [f[0] for f in ct._fields_] == structs[name], name
        for f in structs[name]:
            assert getattr(ct, f).offset == c["%s.%s" % (name, f)], (name, f)
    # the enumerations the result record is indexed with
    assert list(ev.COUNT_NAMES) == ["QR", "QA", "GQ", "DP", "RO", "AO", "RS", "AS", "ASC", "RP", "AP"]
    assert list(ev.TALLY_NAMES) == ["ref_seq", "alt_seq", "alt_clip", "ref_span", "alt_span"]
