"""More libraries in one joint run than an evidence record's 8-bit library index can name (the reference's `-B` list is
unbounded: svtyper/classic.py:145-158, read group -> library parsers.py:432-447): pipeline.library_groups cuts the samples
into groups that fit one device batch each and puts the result records back site-major over all samples.  CPU: the
grouping itself against the one-batch run (cap lowered, oracle engine).  GPU: 130 samples x 2 libraries x 200 sites
through the HIP engine against the oracle, bit-exact."""
import io
import json
import os

import numpy as np
import pytest

import test_native_reads as N
from svtyper_amd import classic, pipeline

HERE = os.path.dirname(os.path.abspath(__file__))
VCF_HEAD = [l for l in open(os.path.join(HERE, "data", "example.vcf")) if l.startswith("##")]
BODY = [
    "1\t50000\td1\tN\t<DEL>\t0\t.\tSVTYPE=DEL;SVLEN=-800;END=50800;STR=+-:10;CIPOS=-5,5;CIEND=-5,5\n",
    "1\t90000\tu1\tN\t<DUP>\t0\t.\tSVTYPE=DUP;SVLEN=1500;END=91500;STR=-+:10;CIPOS=0,0;CIEND=0,0\n",
    "1\t120000\ti1\tN\t<INV>\t0\t.\tSVTYPE=INV;SVLEN=3000;END=123000;STR=++:5,--:5;CIPOS=-10,10;CIEND=-10,10\n",
    "1\t150000\tb1_1\tN\tN]2:40000]\t0\t.\tSVTYPE=BND;STR=++:7;CIPOS=-2,2;CIEND=-2,2;MATEID=b1_2;EVENT=b1\n",
    "2\t40000\tb1_2\tN\tN]1:150000]\t0\t.\tSVTYPE=BND;STR=++:7;CIPOS=-2,2;CIEND=-2,2;MATEID=b1_1;EVENT=b1;SECONDARY\n",
]


class Sink(io.StringIO):
    def close(self):
        pass


def oracle_engine(batch, flags=0, **kw):
    from oracle import c_oracle
    return c_oracle.genotype_batch(batch, flags=flags)


def cohort(tmp_path, n_samples, n_pairs):
    """n_samples synthetic BAMs (two read-group libraries each, tests/test_native_reads.py::_synthetic_bam) + their library JSON"""
    paths, info = [], {}
    for k in range(n_samples):
        path = str(tmp_path / ("s%03d.bam" % k))
        _, inf = N._synthetic_bam(path, seed=300 + k, n_pairs=n_pairs, sample="smp%03d" % k)
        info.update(inf)
        paths.append(path)
    libs = str(tmp_path / "libs.json")
    with open(libs, "w") as f:
        json.dump(info, f)
    return paths, libs


def vcf_text(reps):
    body = []
    for r in range(reps):      # BND ids of their own per repeat (a repeated id would re-pair the mates)
        body += [l.replace("b1_", "b%d_" % r) for l in BODY]
    return "".join(VCF_HEAD) + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" + "".join(body)


def run(paths, libs, text, engine, **kw):
    out = Sink()
    classic.sv_genotype(",".join(paths), io.StringIO(text), out, 20, 1, 1, 1000000, libs, False, None, None, False, None, 1e10,
                        engine=engine, **kw)
    return [l for l in out.getvalue().split("\n") if not l.startswith("##fileDate=")]


def test_groups_are_consecutive_and_fit():
    class S:
        def __init__(self, n):
            self.lib_dict, self.name = dict.fromkeys(range(n)), "s"
    assert pipeline.library_groups([S(2)] * 130) == [list(range(128)), [128, 129]]
    assert pipeline.library_groups([S(200), S(56), S(1), S(256)]) == [[0, 1], [2], [3]]
    assert pipeline.library_groups([S(1)]) == [[0]]
    with pytest.raises(ValueError):
        pipeline.library_groups([S(257)])


@pytest.mark.parametrize("reader,bulk", [("python", "1"), ("native", "1"), ("native", "0")])
def test_grouped_batches_write_the_bytes_of_the_single_batch(tmp_path, monkeypatch, reader, bulk):
    """6 samples x 2 libraries with the cap lowered to 4 (three device batches) == the same run as one batch, for the Python
    reader, the native reader's bulk route and its per-line route -- sample columns and QUAL over all six samples"""
    paths, libs = cohort(tmp_path, 6, 250)
    text = vcf_text(6)
    monkeypatch.setenv("SVT_BULK_VCF", bulk)
    calls = []

    def counting(batch, flags=0, **kw):
        calls.append((batch.n_units, len(batch.libs)))
        return oracle_engine(batch, flags)

    whole = run(paths, libs, text, counting, reader=reader)
    assert [c[1] for c in calls] == [12]
    del calls[:]
    monkeypatch.setattr(pipeline, "MAX_BATCH_LIBS", 4)
    grouped = run(paths, libs, text, counting, reader=reader)
    assert [c[1] for c in calls] == [4, 4, 4] and sum(c[0] for c in calls) == 6 * 24
    assert grouped == whole
    assert sum(1 for l in whole if "\t0/1:" in l or "\t1/1:" in l) > 0


@pytest.mark.gpu
def test_130_samples_of_two_libraries_against_the_oracle(tmp_path, hip_device):
    """260 libraries in one joint run (the reference takes any number): two device batches (128 + 2 samples), every result
    record of the HIP engine bit-identical to the oracle's on the same batch, the VCF identical to the oracle engine's and to
    the per-line route's"""
    from test_hip_parity import assert_parity
    paths, libs = cohort(tmp_path, 130, 120)
    text = vcf_text(50)                       # 200 sites (50 x DEL, DUP, INV, one BND pair)
    real = pipeline.HipEngine(hip_device)
    seen = []

    class Checked:
        """the HIP engine on plain site-major batches (no `accepts_sample_major`), each one checked against the oracle"""

        def __call__(self, batch, flags=0, **kw):
            got = real(batch, flags)
            want = oracle_engine(batch, flags)
            assert_parity(got, want)
            seen.append((batch.n_units, len(batch.libs)))
            return got

    checked = run(paths, libs, text, Checked())
    assert seen == [(200 * 128, 256), (200 * 2, 4)]
    assert len([l for l in checked if l and not l.startswith("#")]) == 250
    assert all(len(l.split("\t")) == 9 + 130 for l in checked if l and not l.startswith("#"))
    default = run(paths, libs, text, None)                     # the default engine: the readers' segments, sample-major
    assert default == checked
    by_oracle = run(paths, libs, text, oracle_engine)
    assert by_oracle == checked
    os.environ["SVT_BULK_VCF"] = "0"
    try:
        per_line = run(paths, libs, text, None)
    finally:
        del os.environ["SVT_BULK_VCF"]
    assert per_line == checked
