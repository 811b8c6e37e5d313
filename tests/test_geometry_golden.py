"""Host geometry layer (svtyper_amd.fragments + packer) against the golden records that
tests/golden/make_golden.py produced with the REFERENCE's fragment objects and predicates."""
import os

import numpy as np

import fakereads
import goldenio as gio
from svtyper_amd import fragments as fr
from svtyper_amd import packer


class _Lib:
    def __init__(self, name, mean, sd):
        self.name, self.mean, self.sd = name, mean, sd


def _pack(site, libs, rg_to_lib, lib_index):
    frags = {}
    for t in site["reads"]:
        r = fakereads.FakeRead(*t)
        lib = rg_to_lib[r.get_tag("RG")]
        if r.query_name in frags:
            frags[r.query_name].add_read(r)
        else:
            frags[r.query_name] = fr.SamFragment(r, lib)
    return packer.pack_fragments(frags, site["breakpoint"], lib_index, 20, 3)


def test_fake_sites_records_match_reference_geometry():
    g = gio.load("fake_sites.json.gz")
    n_rec = n_split = n_cont = n_alt = 0
    for grp in g["groups"]:
        libs = [_Lib(L["name"], gio.fh(L["mean"]), gio.fh(L["sd"])) for L in grp["libraries"]]
        rg_to_lib = {rg: lib for lib, L in zip(libs, grp["libraries"]) for rg in L["readgroups"]}
        lib_index = {id(lib): i for i, lib in enumerate(libs)}
        for site in grp["sites"]:
            got = _pack(site, libs, rg_to_lib, lib_index)
            want = gio.records_from_rows(site["records"])
            assert got.shape == want.shape, site["breakpoint"]["id"]
            for name in want.dtype.names:
                assert np.array_equal(got[name], want[name]), (site["breakpoint"]["id"], name)
            n_rec += len(want)
            n_split += int(((want["seq_l"] | want["seq_r"] | want["clip_l"] | want["clip_r"]) > 0).sum())
            n_cont += int(((want["flags"] & 8) != 0).sum())
            n_alt += int((want["flags"] & 1).sum())
    # the fake sites must exercise the interesting paths
    assert n_rec > 5000 and n_split > 300 and n_cont > 5 and n_alt > 300, (n_rec, n_split, n_cont, n_alt)


def test_cigar_units_like_the_reference_suite():
    """Same literals as the reference's TestCigarParsing (tests/test_svtyper.py:13-56)."""
    assert fr.SplitRead.cigarstring_to_tuple("5H3S2D1N5M3I2P2X1=") == [
        (5, 5), (4, 3), (2, 2), (3, 1), (0, 5), (1, 3), (6, 2), (8, 2), (7, 1)]
    cigar = fr.SplitRead.cigarstring_to_tuple("2S3M1D2M2I3M3S")
    q = fr.SplitRead.SplitPiece.get_query_pos_from_cigar(cigar, True)
    assert (q.query_start, q.query_end, q.query_length) == (3, 13, 15)
    q = fr.SplitRead.SplitPiece.get_query_pos_from_cigar(cigar, False)
    assert (q.query_start, q.query_end, q.query_length) == (2, 12, 15)
    assert fr.SplitRead.get_reference_end_from_cigar(1, fr.SplitRead.cigarstring_to_tuple("2S5M3D2M3S")) == 11
    c = fr.SplitRead.cigarstring_to_tuple("2S5M3D1I1M3S")
    assert fr.SplitRead.get_start_diagonal(fr.SplitRead.SplitPiece(1, 25, True, c, 60)) == 23
    assert fr.SplitRead.get_start_diagonal(fr.SplitRead.SplitPiece(1, 25, False, c, 60)) == 23
    c = fr.SplitRead.cigarstring_to_tuple("2S5M3D2I1M3S")
    for rev in (True, False):
        p = fr.SplitRead.SplitPiece(1, 25, rev, c, 60)
        p.set_reference_end(34)
        assert fr.SplitRead.get_end_diagonal(p) == 34 - (2 + 8)
