"""GPU: the geometry stage (svt_geometry_kernel, include/svtyper_hip.h: svt_fragment) against the
evidence records the REFERENCE's own predicates produced (tests/golden/fake_sites.json.gz and the 211
fixture sites), and both drivers end to end with geometry="device"."""
import os

import numpy as np
import pytest

import fakereads
import goldenio as gio
from svtyper_amd import evidence as ev
from svtyper_amd import fragments as fr
from svtyper_amd import geometry as geo
from svtyper_amd.results import result_from_record

pytestmark = pytest.mark.gpu

CHROMS = {"1": 0, "2": 1}


class _Lib:
    def __init__(self, name, mean, sd):
        self.name, self.mean, self.sd = name, mean, sd


def _fragment_batch(grp):
    libs_json = grp["libraries"]
    libs = [_Lib(L["name"], gio.fh(L["mean"]), gio.fh(L["sd"])) for L in libs_json]
    rg_to_lib = {rg: lib for lib, L in zip(libs, libs_json) for rg in L["readgroups"]}
    lib_index = {id(lib): i for i, lib in enumerate(libs)}
    tid_of = lambda c: CHROMS.get(c, -1)
    b = geo.FragmentBatchBuilder(gio.libraries(libs_json), 1.0, 1.0, 20, 3)
    for site in grp["sites"]:
        frags = {}
        for t in site["reads"]:
            r = fakereads.FakeRead(*t)
            lib = rg_to_lib[r.get_tag("RG")]
            if r.query_name in frags:
                frags[r.query_name].add_read(r)
            else:
                frags[r.query_name] = fr.SamFragment(r, lib)
        b.add(geo.breakpoint_record(site["breakpoint"], tid_of),
              geo.summarise_fragments(frags, site["breakpoint"], lib_index, tid_of))
    return b.build()


def test_fake_sites_device_geometry_matches_reference_records(hip_device):
    from svtyper_amd import hip
    g = gio.load("fake_sites.json.gz")
    n_rec = 0
    for grp in g["groups"]:
        fb = _fragment_batch(grp)
        with hip.DeviceBatch.from_fragments(fb, hip_device, ev.FLAG_SSO_ASSOCIATION, return_records=True) as d:
            want = np.concatenate([gio.records_from_rows(s["records"]) for s in grp["sites"]])
            assert d.records.shape == want.shape
            for name in want.dtype.names:
                bad = np.nonzero(d.records[name] != want[name])[0]
                assert bad.size == 0, (name, bad[:5], d.records[name][bad[:5]], want[name][bad[:5]])
            n_rec += len(want)
            # ... and the likelihood stage fed from those device-resident records
            d.genotype()
            got = d.results()
            for k, s in enumerate(grp["sites"]):
                gio.assert_result_equal(result_from_record(got.rec[k]), gio.golden_result(s["result"]), 1e-6,
                                        s["breakpoint"]["id"])
    assert n_rec > 5000


HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")


def test_fixture_sites_device_geometry(hip_device):
    """Real reads: the 211 fixture breakpoints, fragments built from the BAM by the host layer,
    predicates on the device; records must equal the ones derived with the reference's predicates."""
    import json
    from svtyper_amd import bam, hip, library, pipeline, singlesample, vcf as vcfmod
    g = gio.load("fixture_sites.json.gz")
    sample = library.Sample.from_lib_info(bam.AlignmentFile(os.path.join(DATA, "NA12878.target_loci.sorted.bam")),
                                          json.load(open(os.path.join(DATA, "NA12878.bam.json"))), 1e-3)
    coll = pipeline.UnitCollector([sample], 1, 1, 20, geometry="device")
    for s in g["sites"]:
        frags, many = singlesample.gather_reads(sample, s["breakpoint"], 1000)
        assert not many
        coll.add(s["breakpoint"], 0, frags)
    fb = coll.builders[0].build()
    with hip.DeviceBatch.from_fragments(fb, hip_device, 0, return_records=True) as d:
        want = np.concatenate([gio.records_from_rows(s["records"]) for s in g["sites"]])
        assert d.records.shape == want.shape
        for name in want.dtype.names:
            assert np.array_equal(d.records[name], want[name]), name


@pytest.mark.parametrize("reader", ["python", "native"])
@pytest.mark.parametrize("driver", ["classic", "sso"])
def test_drivers_with_device_geometry(tmp_path, hip_device, driver, reader):
    """example.gt.vcf byte for byte with the predicates on the device, fragments from the Python reader
    or from the native C++ reader + summariser."""
    import test_host_pipeline as T
    out = str(tmp_path / "out.vcf")
    if driver == "classic":
        with open(T.IN_VCF) as inf, open(out, "w") as outf:
            T.classic.sv_genotype(T.IN_BAM, inf, outf, 20, 1, 1, 1000000, T.LIB_JSON, False, None, None, False, None,
                                  1e10, geometry="device", reader=reader)
    else:
        with open(T.IN_VCF) as inf, open(out, "w") as outf:
            T.singlesample.sso_genotype(T.IN_BAM, inf, outf, 20, 1, 1, 1000000, T.LIB_JSON, False, None, False, 1000,
                                        1e10, None, 1000, geometry="device", reader=reader)
    T.same_vcf(T.EXPECTED, out)


def test_native_reader_two_bams_sum_quals(tmp_path, hip_device):
    """Multi-sample interleave of the native collector + QUAL accumulation, against the reference's output."""
    import gzip
    import test_host_pipeline as T
    out = str(tmp_path / "out.vcf")
    with open(T.IN_VCF) as inf, open(out, "w") as outf:
        T.classic.sv_genotype(T.IN_BAM + "," + T.IN_BAM, inf, outf, 20, 1, 1, 1000000, T.LIB_JSON, False, None, None,
                              True, None, 1e10, reader="native")
    want = gzip.open(os.path.join(HERE, "golden", "example.twice.sumquals.gt.vcf.gz"), "rt").read().split("\n")
    got = [l for l in open(out).read().split("\n") if not l.startswith("##fileDate=")]
    assert got == want


@pytest.mark.parametrize("module,extra", [("svtyper_amd.classic", []),
                                          ("svtyper_amd.singlesample", ["--max_reads", "1000"]),
                                          ("svtyper_amd.singlesample", ["--max_reads", "1000", "--reader", "native"]),
                                          ("svtyper_amd.classic", ["--reader", "python", "--geometry", "device"])])
def test_command_line(tmp_path, hip_device, module, extra):
    """the two console entry points, as a user runs them (svtyper / svtyper-sso argument surface)"""
    import subprocess
    import sys
    import test_host_pipeline as T
    out = str(tmp_path / "cli.vcf")
    cmd = [sys.executable, "-m", module, "-B", T.IN_BAM, "-i", T.IN_VCF, "-l", T.LIB_JSON, "-o", out] + extra
    r = subprocess.run(cmd, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    T.same_vcf(T.EXPECTED, out)


def _synthetic_case(tmp_path):
    """BAM from tests/test_native_reads.py::_synthetic_bam + a VCF with its four sites (DEL, DUP, INV, BND pair)"""
    import json
    import test_host_pipeline as T
    import test_native_reads as N
    bam_path = str(tmp_path / "syn.bam")
    _, info = N._synthetic_bam(bam_path, seed=21, n_pairs=900)
    lib_json = str(tmp_path / "syn.json")
    with open(lib_json, "w") as f:
        json.dump(info, f)
    header = [l for l in open(T.IN_VCF) if l.startswith("##")]
    cols = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    body = [
        "1\t50000\td1\tN\t<DEL>\t0\t.\tSVTYPE=DEL;SVLEN=-800;END=50800;STR=+-:10;CIPOS=-5,5;CIEND=-5,5\n",
        "1\t90000\tu1\tN\t<DUP>\t0\t.\tSVTYPE=DUP;SVLEN=1500;END=91500;STR=-+:10;CIPOS=0,0;CIEND=0,0\n",
        "1\t120000\ti1\tN\t<INV>\t0\t.\tSVTYPE=INV;SVLEN=3000;END=123000;STR=++:5,--:5;CIPOS=-10,10;CIEND=-10,10\n",
        "1\t150000\tb1_1\tN\tN]2:40000]\t0\t.\tSVTYPE=BND;STR=++:7;CIPOS=-2,2;CIEND=-2,2;MATEID=b1_2;EVENT=b1\n",
        "2\t40000\tb1_2\tN\tN]1:150000]\t0\t.\tSVTYPE=BND;STR=++:7;CIPOS=-2,2;CIEND=-2,2;MATEID=b1_1;EVENT=b1;SECONDARY\n",
    ]
    vcf_path = str(tmp_path / "syn.vcf")
    with open(vcf_path, "w") as f:
        f.write("".join(header) + cols + "".join(body))
    return bam_path, vcf_path, lib_json


@pytest.mark.parametrize("driver", ["classic", "sso"])
def test_synthetic_bam_all_paths_agree(tmp_path, hip_device, driver):
    """Unusual records (hard clips, N gaps, several SA entries, flags) through every host configuration:
    the CPU oracle engine behind the Python reader is the expectation; the HIP paths (host geometry, device
    geometry, native reader) must write the same bytes."""
    import test_host_pipeline as T
    bam_path, vcf_path, lib_json = _synthetic_case(tmp_path)

    def run(out, **kw):
        with open(vcf_path) as inf, open(out, "w") as outf:
            if driver == "classic":
                T.classic.sv_genotype(bam_path, inf, outf, 20, 1, 1, 1000000, lib_json, False, None, None, False, None,
                                      1e10, **kw)
            else:
                T.singlesample.sso_genotype(bam_path, inf, outf, 20, 1, 1, 1000000, lib_json, False, None, False, 1000,
                                            1e10, None, 1000, **kw)
        return open(out).read()

    want = run(str(tmp_path / "oracle.vcf"), engine=T.oracle_engine)
    assert want.count("\n") > 40 and "\t0/1:" in want or "\t1/1:" in want or "\t0/0:" in want
    for name, kw in (("host", {}), ("device", dict(geometry="device")), ("native", dict(geometry="device", reader="native")),
                     ("native_records", dict(reader="native"))):
        got = run(str(tmp_path / (name + ".vcf")), **kw)
        assert got == want, "%s path differs from the oracle-engine output" % name


def test_library_statistics_from_bam_native_reader(tmp_path, hip_device):
    """No library JSON: both readers derive the libraries from the BAM itself (Library.from_bam; the native
    one through svt_bam_scan_library), write the same JSON and the same genotypes."""
    import test_host_pipeline as T
    outs, jsons = [], []
    for name, kw in (("python", {}), ("native", dict(geometry="device", reader="native"))):
        out, lib_json = str(tmp_path / (name + ".vcf")), str(tmp_path / (name + ".json"))
        with open(T.IN_VCF) as inf, open(out, "w") as outf:
            T.classic.sv_genotype(T.IN_BAM, inf, outf, 20, 1, 1, 1000000, lib_json, False, None, None, False, None,
                                  1e10, **kw)
        outs.append(open(out).read())
        jsons.append(open(lib_json).read())
    assert jsons[0] == jsons[1] and '"histogram"' in jsons[0]
    assert outs[0] == outs[1] and outs[0].count("\n") > 200
