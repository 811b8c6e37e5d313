"""GPU: the reference's inner operator seams BY NAME --
`singlesample.tally_variant_read_fragments(split_slop, min_aligned, breakpoint, sam_fragments, debug) -> counts` and
`singlesample.bayesian_genotype(breakpoint, counts, split_weight, disc_weight, debug) -> result`
(svtyper/singlesample.py:355-404, 406-473) -- against the `counts` / `result` dicts the imported reference produced
for the 211 fixture sites and the 420 fake-read sites (tests/golden/make_golden.py).  counts bit-exact, every FORMAT
value exact, SQ / qual exact too (taken from the bit-exact GL with the host libm)."""
import json
import os

import numpy as np
import pytest

import fakereads
import goldenio as gio
from svtyper_amd import fragments as fr

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


class _Lib:
    def __init__(self, L):
        self.name, self.mean, self.sd = L["name"], gio.fh(L["mean"]), gio.fh(L["sd"])
        self.hist = {int(k): int(v) for k, v in L["hist"].items()}


def _check_site(ss, site, frags):
    bp = site["breakpoint"]
    counts = ss.tally_variant_read_fragments(3, 20, bp, frags, False)
    assert set(counts) == {"ref_seq", "alt_seq", "ref_span", "alt_span", "alt_clip"}
    for t in gio.TALLIES:
        assert float(counts[t]).hex() == site["tallies_sso"][t], (bp["id"], t, counts[t])
    want = gio.golden_result(site["result"])
    if sum(counts.values()) == 0:       # the reference's callers take the blank result here (singlesample.py:492-494)
        got = ss.blank_genotype_result()
    else:
        got = ss.bayesian_genotype(bp, counts, 1, 1, False)
    gio.assert_result_equal(got, want, 0.0, bp["id"])
    return got


def test_seams_on_the_fake_read_sites(hip_device):
    from svtyper_amd import singlesample as ss
    g = gio.load("fake_sites.json.gz")
    n = called = 0
    for grp in g["groups"]:
        libs = [_Lib(L) for L in grp["libraries"]]
        rg_to_lib = {rg: lib for lib, L in zip(libs, grp["libraries"]) for rg in L["readgroups"]}
        for site in grp["sites"]:
            frags = {}
            for t in site["reads"]:
                r = fakereads.FakeRead(*t)
                if r.query_name in frags:
                    frags[r.query_name].add_read(r)
                else:
                    frags[r.query_name] = fr.SamFragment(r, rg_to_lib[r.get_tag("RG")])
            got = _check_site(ss, site, frags)
            n += 1
            called += got["formats"]["GT"] != "./."
    assert n == 420 and called > 300


def test_seams_on_the_fixture_sites(hip_device):
    from svtyper_amd import singlesample as ss
    from svtyper_amd.bam import open_alignment_file
    from svtyper_amd.library import setup_sample
    g = gio.load("fixture_sites.json.gz")
    with open(os.path.join(DATA, "NA12878.bam.json")) as f:
        lib_info = json.load(f)
    sample = setup_sample(open_alignment_file(os.path.join(DATA, "NA12878.target_loci.sorted.bam")), lib_info, 1000000)
    gts = set()
    for site in g["sites"]:
        frags, many = ss.gather_reads(sample, site["breakpoint"], 1000)
        assert not many and len(frags) == site["n_fragments"]
        gts.add(_check_site(ss, site, frags)["formats"]["GT"])
    assert len(g["sites"]) == 211 and gts >= {"0/0", "0/1", "1/1"}


def test_bayesian_genotype_takes_the_counts_as_they_are(hip_device):
    """No zeroing rule and no blank shortcut inside bayesian_genotype (they live in its callers): all-zero counts
    are genotyped -- bayes_gt(0, 0) -> 0/0, GQ 0, SQ 4.771212547196624, GL '0,0,0'  -- and counts
    the zeroing rules would have cleared are used as given."""
    from svtyper_amd import singlesample as ss
    zero = {"ref_seq": 0, "alt_seq": 0, "alt_clip": 0, "ref_span": 0, "alt_span": 0}
    r = ss.bayesian_genotype({"id": "x", "svtype": "DEL"}, zero, 1, 1, False)
    assert r["formats"]["GT"] == "0/0" and r["formats"]["GQ"] == 0 and r["formats"]["GL"] == "0,0,0"
    assert r["formats"]["SQ"] == 4.771212547196624 and r["qual"] == 4.771212547196624 and r["formats"]["AB"] == "."
    only_clip = dict(zero, alt_clip=3.0)          # tally_variant_read_fragments would have zeroed alt_clip
    r = ss.bayesian_genotype({"id": "x", "svtype": "DEL"}, only_clip, 1, 1, False)
    assert r["formats"]["QA"] == 3 and r["formats"]["ASC"] == 3 and r["formats"]["AO"] == 3 and r["formats"]["GT"] == "1/1"
    dup = ss.bayesian_genotype({"id": "x", "svtype": "DUP"}, dict(zero, alt_span=679.0), 1, 1, False)
    assert dup["formats"]["GT"] == "./." and dup["formats"]["GQ"] == "." and dup["formats"]["SQ"] == "."
    weighted = ss.bayesian_genotype({"id": "x", "svtype": "INV"}, dict(zero, ref_seq=10.9, alt_span=5.9), 0.5, 2.0, False)
    assert weighted["formats"]["QR"] == int(0.5 * 10.9) and weighted["formats"]["QA"] == int(2.0 * 5.9)


def test_empty_fragment_dict_gives_the_integer_zero_counts(hip_device):
    from svtyper_amd import singlesample as ss
    bp = {"id": "e", "svtype": "DEL", "var_length": 100, "A": {"chrom": "1", "pos": 10, "ci": [0, 0], "is_reverse": False},
          "B": {"chrom": "1", "pos": 110, "ci": [0, 0], "is_reverse": True}}
    counts = ss.tally_variant_read_fragments(3, 20, bp, {}, False)
    assert counts == {"ref_seq": 0, "alt_seq": 0, "ref_span": 0, "alt_span": 0, "alt_clip": 0}
