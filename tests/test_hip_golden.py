"""GPU: the HIP path against the committed golden vectors (reference-generated), through the C ABI."""
import pytest

import goldenio as gio
from svtyper_amd import evidence as ev
from svtyper_amd.results import result_from_record

pytestmark = pytest.mark.gpu

SQ_TOL = 1e-6  # north_star tolerance; everything else must be exact


def _check(sites, libs_json, flags, hip_device, form="records"):
    """form: "records" = the canonical records through svt_genotype; "packed" = the same sites as packed evidence
    through svt_genotype_packed (several libraries: library switches in the pair streams)"""
    from svtyper_amd import hip
    batch = gio.batch_from_sites(sites, libs_json)
    if form == "packed":
        packed = hip.PackedEvidence.try_pack(batch)
        if packed is None:
            return False
        with packed:
            got = hip.genotype_packed(packed, device=hip_device, flags=flags)
    else:
        got = hip.genotype_batch(batch, device=hip_device, flags=flags)
    for k, s in enumerate(sites):
        for j, t in enumerate(gio.TALLIES):
            if flags & ev.FLAG_SSO_ASSOCIATION:
                want = gio.fh(s["tallies_sso"][t])
            else:
                want = gio.apply_zeroing({x: gio.fh(s["tallies_classic_raw"][x]) for x in gio.TALLIES})[t]
            assert float(got.tallies[k, j]).hex() == float(want).hex(), (s["breakpoint"]["id"], t)
        gio.assert_result_equal(result_from_record(got.rec[k]), gio.golden_result(s["result"]), SQ_TOL,
                                s["breakpoint"]["id"])
    return True


@pytest.mark.parametrize("form", ["records", "packed"])
def test_fixture_sites_sso(hip_device, form):
    g = gio.load("fixture_sites.json.gz")
    assert _check(g["sites"], g["libraries"], ev.FLAG_SSO_ASSOCIATION, hip_device, form)


@pytest.mark.parametrize("form", ["records", "packed"])
def test_fixture_sites_classic(hip_device, form):
    g = gio.load("fixture_sites.json.gz")
    assert _check(g["sites"], g["libraries"], 0, hip_device, form)


@pytest.mark.parametrize("form", ["records", "packed"])
def test_fake_sites(hip_device, form):
    g = gio.load("fake_sites.json.gz")
    ran = [_check(grp["sites"], grp["libraries"], ev.FLAG_SSO_ASSOCIATION, hip_device, form) for grp in g["groups"]]
    # packed: the three-library group too (library switches); a group whose library geometry is outside the format's range is declined
    multi = [len(grp["libraries"]) > 1 for grp in g["groups"]]
    assert any(r and m for r, m in zip(ran, multi)) and (form == "packed" or all(ran))


def test_bayes_gt_seam(hip_device):
    """statistics.bayes_gt / log_choose through svt_bayes_gt: bit-exact against the reference's values
    (bayes_grid golden + SURVEY.md 8c-iii known answers)."""
    import numpy as np
    from svtyper_amd import statistics as st
    assert st.bayes_gt(10, 5, False) == (-11.526789785541196, -1.0378946027607374, -6.751232120604396)
    assert st.bayes_gt(10, 5, True) == (-6.566092721825519, -0.9863948195616774, -0.6689635319561438)
    assert st.bayes_gt(0, 0, True) == (0.0, 0.0, 0.0)
    assert st.log_choose(200, 100) == 58.956881330608695
    g = gio.load("bayes_grid.json.gz")
    ref, alt, dup, want = [], [], [], []
    for case in g["cases"]:
        f = gio.golden_result(case["result"])["formats"]
        ref.append(f["QR"]); alt.append(f["QA"]); dup.append(case["svtype"] == "DUP")
        want.append([gio.fh(x) for x in case["gl"]])
    got = st.bayes_gt_array(ref, alt, dup)
    assert np.array_equal(got.view(np.uint64), np.asarray(want).view(np.uint64))
