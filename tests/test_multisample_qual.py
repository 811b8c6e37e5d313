"""Multi-sample QUAL (svtyper/classic.py:216-217,485,498) against the imported reference beyond the two-BAM fixture:
three synthetic BAMs whose MIDDLE sample has no read at two of the sites -- its blank result resets the running
QUAL there -- with and without --sum_quals and with incoming QUALs that are not 0.  Goldens:
tests/golden/three.sumquals.gt.vcf.gz / three.gt.vcf.gz (tests/golden/make_golden.py: make_three_sample_vcf, which
regenerates the very same input files through three_sample_case below)."""
import gzip
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def three_sample_case(workdir):
    """Three synthetic BAMs (tests/test_native_reads.py::_synthetic_bam; the middle sample has NO read at the DUP and
    the BND site, so its result there is the blank one), their library JSON and a VCF whose incoming QUALs are not 0.
    Deterministic: the test regenerates the very same files (tests/test_multisample_qual.py)."""
    import test_native_reads as N
    import test_host_pipeline as T
    info = {}
    bams = []
    for name, seed, pairs, only in (("left", 71, 800, None), ("mid", 72, 500, (0, 2)), ("right", 73, 900, None)):
        path = os.path.join(workdir, name + ".bam")
        _, inf = N._synthetic_bam(path, seed=seed, n_pairs=pairs, sample=name, only_sites=only)
        info.update(inf)
        bams.append(path)
    lib_json = os.path.join(workdir, "three.json")
    with open(lib_json, "w") as f:
        json.dump(info, f)
    header = [l for l in open(T.IN_VCF) if l.startswith("##")]
    cols = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    body = [
        "1\t50000\td1\tN\t<DEL>\t12.5\t.\tSVTYPE=DEL;SVLEN=-800;END=50800;STR=+-:10;CIPOS=-5,5;CIEND=-5,5\n",
        "1\t90000\tu1\tN\t<DUP>\t7\t.\tSVTYPE=DUP;SVLEN=1500;END=91500;STR=-+:10;CIPOS=0,0;CIEND=0,0\n",
        "1\t120000\ti1\tN\t<INV>\t0\t.\tSVTYPE=INV;SVLEN=3000;END=123000;STR=++:5,--:5;CIPOS=-10,10;CIEND=-10,10\n",
        "1\t150000\tb1_1\tN\tN]2:40000]\t3.25\t.\tSVTYPE=BND;STR=++:7;CIPOS=-2,2;CIEND=-2,2;MATEID=b1_2;EVENT=b1\n",
        "2\t40000\tb1_2\tN\tN]1:150000]\t3.25\t.\tSVTYPE=BND;STR=++:7;CIPOS=-2,2;CIEND=-2,2;MATEID=b1_1;EVENT=b1;SECONDARY\n",
    ]
    vcf_path = os.path.join(workdir, "three.vcf")
    with open(vcf_path, "w") as f:
        f.write("".join(header) + cols + "".join(body))
    return ",".join(bams), vcf_path, lib_json


def _run(tmp_path, name, sum_quals, **kw):
    from svtyper_amd import classic
    bams, vcf_path, lib_json = three_sample_case(str(tmp_path))
    out = str(tmp_path / (name + ".vcf"))
    with open(vcf_path) as inf, open(out, "w") as outf:
        classic.sv_genotype(bams, inf, outf, 20, 1, 1, 1000000, lib_json, False, None, None, sum_quals, None, 1e10, **kw)
    return [l for l in open(out).read().split("\n") if not l.startswith("##fileDate=")]


def _golden(sum_quals):
    name = "three.sumquals.gt.vcf.gz" if sum_quals else "three.gt.vcf.gz"
    return gzip.open(os.path.join(HERE, "golden", name), "rt").read().split("\n")


def _same(got, want):
    assert len(got) == len(want)
    for i, (x, y) in enumerate(zip(got, want)):
        assert x == y, "line %d\n%s\n%s" % (i + 1, x, y)


@pytest.mark.parametrize("sum_quals", [True, False])
def test_three_samples_blank_in_the_middle_oracle_engine(tmp_path, sum_quals):
    import test_host_pipeline as T
    want = _golden(sum_quals)
    body = [l.split("\t") for l in want if l and not l.startswith("#")]
    assert len(body) == 5 and all(len(r) == 12 for r in body)
    assert sum(r[10].startswith("./.:.:0:0:0") for r in body) == 3       # the middle sample is blank at u1 and the BND pair
    _same(_run(tmp_path, "oracle", sum_quals, engine=T.oracle_engine), want)
    # ... and with the native reader handing evidence records of the three BAMs to the same engine (units interleaved
    # site-major over the samples by NativeUnitCollector._run_records)
    _same(_run(tmp_path, "oracle_native", sum_quals, engine=T.oracle_engine, reader="native"), want)

    # ... and handed over SAMPLE-major, as the HIP engine takes them (accepts_sample_major: the readers' arrays concatenated,
    # the device writes the result records site-major).  The stand-in puts the units in site-major order itself.
    class SampleMajorOracle:
        accepts_sample_major = True
        supports_site_qual = True
        calls = 0

        def __call__(self, batch, flags=0, site_qual=None, sample_major=0):
            import numpy as np
            from svtyper_amd import hip, synth
            from svtyper_amd.evidence import SegmentedBatch
            assert isinstance(batch, SegmentedBatch) and len(batch.segments) == 3
            batch = batch.joined()
            assert sample_major == 3 and batch.n_units % 3 == 0
            assert (batch.units["sample"] == np.repeat(np.arange(3), batch.n_units // 3)).all()
            n_sites = batch.n_units // 3
            site_major = synth.permute_units(batch, (np.arange(batch.n_units) % 3) * n_sites + np.arange(batch.n_units) // 3)
            res = T.oracle_engine(site_major, flags)
            if site_qual is not None:
                res.site_qual = hip.site_qual_host(res, site_qual[0], site_qual[1])
            SampleMajorOracle.calls += 1
            return res
    _same(_run(tmp_path, "oracle_native_sample_major", sum_quals, engine=SampleMajorOracle(), reader="native"), want)
    assert SampleMajorOracle.calls >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("sum_quals", [True, False])
@pytest.mark.parametrize("kw", [{}, {"geometry": "device"}, {"geometry": "device", "reader": "native"}, {"reader": "native"}],
                         ids=["host", "device-geometry", "native-reader", "native-reader-own-geometry"])
def test_three_samples_blank_in_the_middle_hip(tmp_path, hip_device, sum_quals, kw):
    _same(_run(tmp_path, "hip", sum_quals, **kw), _golden(sum_quals))
