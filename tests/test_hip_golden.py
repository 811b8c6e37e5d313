"""GPU: the HIP path against the committed golden vectors (reference-generated), through the C ABI."""
import pytest

import goldenio as gio
from svtyper_amd import evidence as ev
from svtyper_amd.results import result_from_record

pytestmark = pytest.mark.gpu

SQ_TOL = 1e-6  # north_star tolerance; everything else must be exact


def _check(sites, libs_json, flags, hip_device):
    from svtyper_amd import hip
    batch = gio.batch_from_sites(sites, libs_json)
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


def test_fixture_sites_sso(hip_device):
    g = gio.load("fixture_sites.json.gz")
    _check(g["sites"], g["libraries"], ev.FLAG_SSO_ASSOCIATION, hip_device)


def test_fixture_sites_classic(hip_device):
    g = gio.load("fixture_sites.json.gz")
    _check(g["sites"], g["libraries"], 0, hip_device)


def test_fake_sites(hip_device):
    g = gio.load("fake_sites.json.gz")
    for grp in g["groups"]:
        _check(grp["sites"], grp["libraries"], ev.FLAG_SSO_ASSOCIATION, hip_device)
