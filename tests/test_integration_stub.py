"""The reference-side binding printed in INTEGRATION.md section 2 (`svtyper/hipbackend.py`, what a maintainer of
hall-lab/svtyper would add) is EXECUTED here, so the document cannot rot: its code block is cut out of the markdown,
run against the in-tree libsvtyper_hip.so and fed the 420 fake-read sites -- fragment objects that answer the
reference's predicates (svtyper_amd/fragments.py exposes the same ones) -- the way section 2 describes:
pack_fragment per fragment in sorted(query_name) order, unit_header per breakpoint, one genotype_units call.

CPU part: the stub imports (ABI version, signatures), its dtypes are the library's, and its rows equal the records
the imported reference's own predicates produced (tests/golden/fake_sites.json.gz).
GPU part: its results equal the reference's result dicts for those sites (both associations of the split-read sums
are exercised: the goldens hold the singlesample tallies and the classic results)."""
import os
import re

import numpy as np
import pytest

import fakereads
import goldenio as gio
from svtyper_amd import evidence as ev, fragments as fr, hip, results

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    code = [b for b in blocks if b.lstrip().startswith("# svtyper/hipbackend.py")]
    assert len(code) == 1, "INTEGRATION.md must hold exactly one `# svtyper/hipbackend.py` code block"
    src = code[0]
    assert 'C.CDLL("libsvtyper_hip.so")' in src
    src = src.replace('C.CDLL("libsvtyper_hip.so")', "C.CDLL(%r)" % hip.LIB_PATH)   # the in-tree build, not the loader path
    ns = {}
    exec(compile(src, "INTEGRATION.md:hipbackend", "exec"), ns)
    return ns


@pytest.fixture(scope="module")
def stub():
    return _stub()


class _Lib:
    def __init__(self, L):
        self.name, self.mean, self.sd = L["name"], gio.fh(L["mean"]), gio.fh(L["sd"])
        self.hist = {int(k): int(v) for k, v in L["hist"].items()}


def _sites(stub):
    """every fake-read site as (site json, rows from the stub's pack_fragment, unit header from the stub)"""
    g = gio.load("fake_sites.json.gz")
    for grp in g["groups"]:
        libs = [_Lib(L) for L in grp["libraries"]]
        rg_to_lib = {rg: lib for lib, L in zip(libs, grp["libraries"]) for rg in L["readgroups"]}
        lib_idx = {id(lib): i for i, lib in enumerate(libs)}
        for site in grp["sites"]:
            frags = {}
            for t in site["reads"]:
                r = fakereads.FakeRead(*t)
                if r.query_name in frags:
                    frags[r.query_name].add_read(r)
                else:
                    frags[r.query_name] = fr.SamFragment(r, rg_to_lib[r.get_tag("RG")])
            rows = []
            for name in sorted(frags):                                   # classic.py:296
                f = frags[name]
                rows += stub["pack_fragment"](f, site["breakpoint"], lib_idx[id(f.lib)], 20, 3)
            yield grp, site, rows, stub["unit_header"](site["breakpoint"])


def test_the_stub_imports_and_its_types_are_the_librarys(stub):
    assert stub["L"].svt_version() == hip.ABI_VERSION
    for mine, theirs in ((stub["REC"], ev.RECORD_DTYPE), (stub["UNIT"], ev.UNIT_DTYPE), (stub["RES"], ev.RESULT_DTYPE)):
        assert mine.itemsize == theirs.itemsize
        assert [mine.fields[n][1] for n in mine.names] == [theirs.fields[n][1] for n in theirs.names]
    import ctypes as C
    assert C.sizeof(stub["Lib"]) == C.sizeof(ev.CLibrary) and C.sizeof(stub["Batch"]) == C.sizeof(ev.CEvidenceBatch)


def test_the_stubs_rows_are_the_references_records(stub):
    n = n_rows = 0
    for grp, site, rows, unit in _sites(stub):
        assert [list(r) for r in rows] == [list(r) for r in site["records"]], site["breakpoint"]["id"]
        want = gio.unit_from_breakpoint(site["breakpoint"])
        assert unit.tobytes() == want.astype(stub["UNIT"]).tobytes()
        n += 1
        n_rows += len(rows)
    assert n == 420 and n_rows > 10000


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [(0,), (0, 0, 0)])
def test_the_stub_genotypes_the_fake_read_sites(stub, hip_device, devices):
    import ctypes as C
    g = gio.load("fake_sites.json.gz")
    per_group = {}
    for grp, site, rows, unit in _sites(stub):
        per_group.setdefault(id(grp), (grp, []))[1].append((site, rows, unit))
    n = called = 0
    for grp, items in per_group.values():
        tables = gio.libraries(grp["libraries"])                      # dense histograms (svt_library.hist)
        libs = (stub["Lib"] * len(tables))()
        for i, t in enumerate(tables):
            libs[i].hist = t.hist.ctypes.data_as(C.POINTER(C.c_uint32))
            libs[i].key_min, libs[i].n_bins, libs[i].mean, libs[i].sd = t.key_min, len(t.hist), t.mean, t.sd
        offs = np.zeros(len(items) + 1, np.uint64)
        offs[1:] = np.cumsum([len(rows) for _, rows, _ in items])
        recs = np.zeros(int(offs[-1]), stub["REC"])
        flat = [r for _, rows, _ in items for r in rows]
        if flat:
            arr = np.asarray(flat, dtype=np.int64)
            for i, name in enumerate(stub["REC"].names):
                recs[name] = arr[:, i]
        units = np.concatenate([u for _, _, u in items])
        out = stub["genotype_units"](offs, units, recs, libs, 1.0, 1.0, sso=False, devices=devices)
        for (site, _, _), rec in zip(items, out.view(ev.RESULT_DTYPE)):
            got = results.result_from_record(rec)
            gio.assert_result_equal(got, gio.golden_result(site["result"]), 0.0, site["breakpoint"]["id"])
            n += 1
            called += got["formats"]["GT"] != "./."
        # the singlesample association: tallies bit-identical to the reference's tally_variant_read_fragments
        out_sso = stub["genotype_units"](offs, units, recs, libs, 1.0, 1.0, sso=True, devices=devices)
        for (site, _, _), rec in zip(items, out_sso.view(ev.RESULT_DTYPE)):
            raw = {t: gio.fh(site["tallies_sso"][t]) for t in gio.TALLIES}
            want = gio.apply_zeroing(raw)
            if sum(raw.values()) > 0:
                for i, t in enumerate(ev.TALLY_NAMES):
                    assert float(rec["tallies"][i]).hex() == float(want[t]).hex(), (site["breakpoint"]["id"], t)
    assert n == 420 and called > 300
