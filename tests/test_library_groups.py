"""Many libraries in one joint run (the reference's `-B` list is unbounded: svtyper/classic.py:145-158, read group -> library
parsers.py:432-447).  ABI 18: an evidence record names its library with sixteen bits and a unit's window hint carries a
16-bit first library, so 130 samples x 2 libraries are ONE device batch of 260 libraries; beyond 65 536 libraries
pipeline.library_groups cuts the samples into groups that fit one batch each and puts the result records back site-major.
CPU: the grouping against the one-batch run (cap lowered, oracle engine).  GPU: 130 samples x 2 libraries x 200 sites
through the HIP engine against the oracle, bit-exact -- as one batch and, cap lowered, as two."""
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


def test_groups_are_consecutive_and_fit(monkeypatch):
    class S:
        def __init__(self, n):
            self.lib_dict, self.name = dict.fromkeys(range(n)), "s"
    assert pipeline.MAX_BATCH_LIBS == 65536
    assert pipeline.library_groups([S(2)] * 130) == [list(range(130))]
    monkeypatch.setattr(pipeline, "MAX_BATCH_LIBS", 256)
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
@pytest.mark.parametrize("cap", [65536, 256])
def test_130_samples_of_two_libraries_against_the_oracle(tmp_path, hip_device, monkeypatch, cap):
    """260 libraries in one joint run (the reference takes any number): ONE device batch of 260 libraries (16-bit library
    index, 16-bit first library in the window hints), or -- the cap lowered to the eight bits of ABI <= 17 -- two batches
    (128 + 2 samples); every result record of the HIP engine bit-identical to the oracle's on the same batch, the VCF
    identical to the oracle engine's and to the per-line route's"""
    monkeypatch.setattr(pipeline, "MAX_BATCH_LIBS", cap)
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
    assert seen == ([(200 * 130, 260)] if cap > 256 else [(200 * 128, 256), (200 * 2, 4)])
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
    # fragment summaries with the geometry predicates on the device (svt_bam_summarise -> svt_batch_create_from_fragments): the
    # library index travels in the summaries' sixteen reserved bits
    assert run(paths, libs, text, None, geometry="device") == checked


@pytest.mark.gpu
def test_one_batch_of_more_than_256_libraries_through_every_mode(hip_device):
    """400 samples x 1-3 libraries (~800 libraries) in ONE batch, straight through the C ABI against the oracle: with the
    window hints (first library up to ~800: sixteen bits), sample-major with site-major results, without hints (windows read
    off the records by svt_window_scan_kernel), and with every table through L2 (general mode: <= 1 024 libraries);
    a record that names a library outside its unit's window, or beyond the batch, is still the contract violation it was"""
    from svtyper_amd import evidence as ev, hip, synth
    from oracle import c_oracle
    from test_hip_parity import assert_parity
    batch = synth.make_multisample(12, 400, seed=77, mean_frags=14, sd_frags=5, min_frags=0, max_frags=30)
    assert 600 < len(batch.libs) <= 1024 and int(ev.unit_libs_first(batch.units["libs"]).max()) > 256
    assert int((batch.records["flags"] >> ev.REC_LIB_SHIFT).max()) > 256
    want = c_oracle.genotype_batch(batch)
    assert_parity(hip.genotype_batch(batch, device=hip_device), want)
    assert_parity(hip.genotype_batch(batch, device=hip_device, flags=ev.FLAG_GENERAL_TABLES), want)
    for flags in (0, ev.FLAG_SSO_ASSOCIATION, ev.FLAG_RESULT96):
        with hip.DeviceBatch(batch, hip_device, flags) as d:
            assert d.table_mode() == 1          # library windows
            d.genotype(sync=True)
            assert_parity(d.results(), c_oracle.genotype_batch(batch, flags=flags & ev.FLAG_SSO_ASSOCIATION))
    bare = synth.permute_units(batch, np.arange(batch.n_units))
    bare.units["libs"] = 0                      # no hints: the scan finds every unit's window
    with hip.DeviceBatch(bare, hip_device, 0) as d:
        assert d.table_mode() == 1
        d.genotype(sync=True)
        assert_parity(d.results(), want)
    by_sample, _ = synth.to_sample_major(batch, 400)
    with hip.DeviceBatch(by_sample, hip_device, 0) as d:
        d.result_order(400)
        d.genotype(sync=True)
        assert d.results().rec.tobytes() == hip.genotype_batch(batch, device=hip_device).rec.tobytes()
    # contract violations
    bad = synth.permute_units(batch, np.arange(batch.n_units))
    u = int(np.nonzero(np.diff(bad.rec_offset.astype(np.int64)) > 0)[0][-1])          # a unit of a late sample that has records
    first = int(ev.unit_libs_first(int(bad.units["libs"][u])))
    bad.units["libs"][u] = ev.unit_libs(first - 5, 1)
    with pytest.raises(hip.SvtyperHipError) as e:
        hip.genotype_batch(bad, device=hip_device)
    assert "lib index" in str(e.value)
    beyond = synth.permute_units(batch, np.arange(batch.n_units))
    r = int(beyond.rec_offset[u])
    beyond.records["flags"][r] = (int(beyond.records["flags"][r]) & 0xFF) | (len(batch.libs) + 3) << ev.REC_LIB_SHIFT
    for b2 in (beyond,):
        with pytest.raises(hip.SvtyperHipError):
            hip.genotype_batch(b2, device=hip_device)
    reserved = synth.permute_units(batch, np.arange(batch.n_units))
    reserved.records["flags"][r] |= 1 << 26
    with pytest.raises(hip.SvtyperHipError) as e:
        hip.genotype_batch(reserved, device=hip_device)
    assert "reserved" in str(e.value) or "undefined" in str(e.value)
    # packed evidence names a library with eight bits: such a batch stays canonical
    assert hip.PackedEvidence.try_pack(batch) is None


def test_more_than_256_libraries_on_the_cpu_side():
    """the oracle (C and Python) and the batch validation read sixteen bits of library index"""
    from svtyper_amd import evidence as ev, synth
    from oracle import c_oracle, py_oracle
    batch = synth.make_multisample(3, 300, seed=5, mean_frags=6, sd_frags=2, min_frags=0, max_frags=12)
    assert len(batch.libs) > 400 and int((batch.records["flags"] >> ev.REC_LIB_SHIFT).max()) > 256
    a, b = c_oracle.genotype_batch(batch), py_oracle.genotype_batch(batch, 0)
    assert np.array_equal(a.gt, b.gt) and np.array_equal(a.counts, b.counts)
    assert np.array_equal(a.gl.view(np.uint64), b.gl.view(np.uint64))
    # the libraries matter: the same records against the FIRST sample's library give other answers somewhere
    moved = synth.permute_units(batch, np.arange(batch.n_units))
    moved.records["flags"] &= 0xFF
    assert not np.array_equal(c_oracle.genotype_batch(moved).gl, a.gl)
