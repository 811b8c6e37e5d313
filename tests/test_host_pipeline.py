"""configs[0]: the reference's own fixture through the host pipeline (VCF model, BAM reader,
fragments, packer, result formatting) -- byte-compared with tests/data/example.gt.vcf exactly as the
reference's integration tests do (tests/test_svtyper.py:66-99, tests/test_singlesample.py:19-69).

CPU run: the likelihood engine seam is filled by the ORACLE (test infrastructure) so that the
plumbing can be checked without a GPU; the gpu-marked twin below runs the very same calls through
the product's HIP engine."""
import os
import re

import pytest

from svtyper_amd import classic, singlesample

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")
IN_VCF = os.path.join(DATA, "example.vcf")
IN_BAM = os.path.join(DATA, "NA12878.target_loci.sorted.bam")
LIB_JSON = os.path.join(DATA, "NA12878.bam.json")
EXPECTED = os.path.join(DATA, "example.gt.vcf")


def oracle_engine(batch, flags=0):
    from oracle import c_oracle
    return c_oracle.genotype_batch(batch, flags=flags)


def same_vcf(a, b):
    """diff -I '^##fileDate='"""
    strip = lambda p: [l for l in open(p).read().split("\n") if not re.match(r"^##fileDate=", l)]
    la, lb = strip(a), strip(b)
    assert len(la) == len(lb), (len(la), len(lb))
    for i, (x, y) in enumerate(zip(la, lb)):
        assert x == y, "line %d differs:\n%s\n%s" % (i + 1, x, y)


def run_classic(out, engine, **kw):
    with open(IN_VCF) as inf, open(out, "w") as outf:
        classic.sv_genotype(bam_string=IN_BAM, vcf_in=inf, vcf_out=outf, min_aligned=20, split_weight=1,
                            disc_weight=1, num_samp=1000000, lib_info_path=LIB_JSON, debug=False,
                            alignment_outpath=None, ref_fasta=None, sum_quals=False, max_reads=None,
                            max_ci_dist=1e10, engine=engine, **kw)


def run_sso(out, engine, cores, **kw):
    with open(IN_VCF) as inf, open(out, "w") as outf:
        singlesample.sso_genotype(bam_string=IN_BAM, vcf_in=inf, vcf_out=outf, min_aligned=20, split_weight=1,
                                  disc_weight=1, num_samp=1000000, lib_info_path=LIB_JSON, debug=False,
                                  ref_fasta=None, sum_quals=False, max_reads=1000, max_ci_dist=1e10, cores=cores,
                                  batch_size=1000, engine=engine, **kw)


def test_classic_integration_oracle_engine(tmp_path):
    out = str(tmp_path / "out.vcf")
    run_classic(out, oracle_engine)
    same_vcf(EXPECTED, out)


@pytest.mark.parametrize("cores", [None, 1])
def test_sso_integration_oracle_engine(tmp_path, cores):
    out = str(tmp_path / "out.vcf")
    run_sso(out, oracle_engine, cores)
    same_vcf(EXPECTED, out)


@pytest.mark.parametrize("driver", ["classic", "sso"])
def test_the_reference_signature_takes_the_native_bulk_route(tmp_path, monkeypatch, driver):
    """A caller with the reference's own arguments (no `reader=`) gets the C++ reader and the bulk VCF route -- no Variant
    object per line -- and the same bytes as the portable Python reader."""
    from svtyper_amd import bulk_vcf, native_reads, pipeline, vcf
    assert pipeline.resolve_reader(None) == "native" and pipeline.resolve_reader("python") == "python"
    made = {"variant": 0, "evidence": 0, "blocks": 0}
    init, evidence, parse = vcf.Variant.__init__, native_reads.NativeBam.evidence, bulk_vcf.VcfParser.parse
    monkeypatch.setattr(vcf.Variant, "__init__", lambda self, *a: (made.__setitem__("variant", made["variant"] + 1), init(self, *a))[1])
    monkeypatch.setattr(native_reads.NativeBam, "evidence", lambda self, *a: (made.__setitem__("evidence", made["evidence"] + 1), evidence(self, *a))[1])
    monkeypatch.setattr(bulk_vcf.VcfParser, "parse", lambda self, *a: (made.__setitem__("blocks", made["blocks"] + 1), parse(self, *a))[1])
    run = (lambda out, **kw: run_classic(out, oracle_engine, **kw)) if driver == "classic" else (lambda out, **kw: run_sso(out, oracle_engine, None, **kw))
    default = str(tmp_path / "default.vcf")
    run(default)
    assert made == {"variant": 0, "evidence": 1, "blocks": 1}
    same_vcf(EXPECTED, default)
    portable = str(tmp_path / "python.vcf")
    run(portable, reader="python")
    assert made["variant"] == 212 and made["evidence"] == 1
    same_vcf(default, portable)


@pytest.mark.parametrize("driver", ["classic", "sso"])
@pytest.mark.parametrize("reader", ["python", None])
def test_small_chunks_through_the_chunk_pipeline(tmp_path, monkeypatch, driver, reader):
    """Many device batches per run: chunk k is genotyped on the worker thread while chunk k+1 is parsed,
    the output order and bytes stay those of the single-batch run (BND mates straddle chunk borders) -- per line with the
    Python reader, in blocks of text on the default route."""
    calls = []

    def counting_engine(batch, flags=0, **kw):
        calls.append(batch.n_units)
        return oracle_engine(batch, flags)

    out = str(tmp_path / "out.vcf")
    monkeypatch.setenv("SVT_BULK_BLOCK_SITES", "17")
    if driver == "classic":
        monkeypatch.setattr(classic, "CHUNK_UNITS", 17)
        run_classic(out, counting_engine, reader=reader)
    else:
        monkeypatch.setattr(singlesample, "CHUNK_UNITS", 17)
        run_sso(out, counting_engine, None, reader=reader)
    same_vcf(EXPECTED, out)
    assert len(calls) >= 3 and max(calls) <= (17 if reader == "python" else 200)


@pytest.mark.parametrize("driver", ["classic", "sso"])
def test_bulk_sample_columns_and_general_path_write_the_same_bytes(tmp_path, monkeypatch, driver):
    """The fixture is a sites-only VCF, so the per-line route takes the bulk formatter (svt_format_results); with it
    switched off the per-sample Genotype path must produce the same file (== the reference's output)."""
    from svtyper_amd import pipeline
    calls = []
    orig = pipeline.SampleColumnWriter.columns
    monkeypatch.setattr(pipeline.SampleColumnWriter, "columns", lambda self, r: (calls.append(r.n_units), orig(self, r))[1])
    run = ((lambda out: run_classic(out, oracle_engine, reader="python")) if driver == "classic" else
           (lambda out: run_sso(out, oracle_engine, None, reader="python")))
    bulk = str(tmp_path / "bulk.vcf")
    run(bulk)
    assert calls, "the bulk formatter was not used"
    same_vcf(EXPECTED, bulk)
    monkeypatch.setattr(pipeline.SampleColumnWriter, "eligible", lambda self, v: False)
    n = len(calls)
    general = str(tmp_path / "general.vcf")
    run(general)
    assert len(calls) == n
    same_vcf(EXPECTED, general)


def test_chunk_pipeline_orders_results_and_surfaces_errors():
    from svtyper_amd.pipeline import ChunkPipeline
    import time
    seen = []
    pipe = ChunkPipeline()
    for k in range(5):
        pipe.submit(lambda k=k: (time.sleep(0.02 * (5 - k)), k)[1], seen.append)
    pipe.close()
    assert seen == [0, 1, 2, 3, 4]
    pipe = ChunkPipeline()
    pipe.submit(lambda: 1 / 0, seen.append)
    with pytest.raises(ZeroDivisionError):
        pipe.close()


def test_default_engine_fails_loudly_without_gpu(tmp_path):
    from svtyper_amd import hip
    hip.load()
    if hip.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hip.SvtyperHipError):
        run_classic(str(tmp_path / "out.vcf"), None)


@pytest.mark.gpu
def test_classic_integration_hip(tmp_path, hip_device):
    out = str(tmp_path / "out.vcf")
    run_classic(out, None)   # default engine = HIP
    same_vcf(EXPECTED, out)


@pytest.mark.gpu
@pytest.mark.parametrize("cores", [None, 1])
def test_sso_integration_hip(tmp_path, hip_device, cores):
    out = str(tmp_path / "out.vcf")
    run_sso(out, None, cores)
    same_vcf(EXPECTED, out)


@pytest.mark.gpu
@pytest.mark.parametrize("driver", ["classic", "sso"])
@pytest.mark.parametrize("geometry,reader", [("host", "python"), ("device", "python"), ("device", "native")])
def test_small_chunks_through_the_chunk_pipeline_hip(tmp_path, monkeypatch, hip_device, driver, geometry, reader):
    """The gpu twin of test_small_chunks_through_the_chunk_pipeline: chunk size << 211 so a run is a dozen device
    batches through the HIP engine -- svt_batch_create / svt_batch_destroy per chunk with the pooled device buffers
    and the pinned ring reused, chunk k on the worker thread while chunk k+1 is parsed -- and the bytes stay the
    reference's."""
    from svtyper_amd import pipeline
    calls = []
    real = pipeline.HipEngine(hip_device)

    class Counting:
        supports_site_qual = True

        def __call__(self, batch, flags=0, **kw):
            calls.append(batch.n_units)
            return real(batch, flags, **kw)

        def genotype_fragments(self, fbatch, flags=0, **kw):
            calls.append(fbatch.n_units)
            return real.genotype_fragments(fbatch, flags, **kw)

    out = str(tmp_path / "out.vcf")
    kw = dict(engine=Counting(), geometry=geometry, reader=reader)
    monkeypatch.setenv("SVT_BULK_BLOCK_SITES", "17")      # (reader="native": blocks of text of about that many lines)
    with open(IN_VCF) as inf, open(out, "w") as outf:
        if driver == "classic":
            monkeypatch.setattr(classic, "CHUNK_UNITS", 17)
            classic.sv_genotype(IN_BAM, inf, outf, 20, 1, 1, 1000000, LIB_JSON, False, None, None, False, None, 1e10, **kw)
        else:
            monkeypatch.setattr(singlesample, "CHUNK_UNITS", 17)
            singlesample.sso_genotype(IN_BAM, inf, outf, 20, 1, 1, 1000000, LIB_JSON, False, None, False, 1000, 1e10,
                                      None, 1000, **kw)
    same_vcf(EXPECTED, out)
    assert len(calls) >= 12 and max(calls) <= (17 if reader == "python" else 40)


def test_library_from_bam_matches_reference():
    """Library.from_bam / Sample.from_bam against the statistics the imported reference computed
    from the same BAM (tests/golden/library_from_bam.json.gz)."""
    import goldenio as gio
    from svtyper_amd import bam, library
    g = gio.load("library_from_bam.json.gz")
    sample = library.Sample.from_bam(bam.AlignmentFile(IN_BAM), 1000000, 1e-3)
    assert sample.name == g["sample"] and sample.active_libs == g["active_libs"]
    assert float(sample.get_fetch_flank(3)).hex() == g["fetch_flank_z3"]
    assert (sample.bam_mapped, sample.bam_unmapped) == (g["mapped"], g["unmapped"])
    assert len(sample.lib_dict) == len(g["libraries"])
    for lib, want in zip(sample.lib_dict.values(), g["libraries"]):
        assert lib.name == want["name"] and lib.readgroups == want["readgroups"]
        assert lib.read_length == want["read_length"]
        assert float(lib.mean).hex() == want["mean"] and float(lib.sd).hex() == want["sd"]
        assert float(lib.prevalence).hex() == want["prevalence"]
        assert {str(k): int(v) for k, v in lib.hist.items()} == want["hist"]


def test_sso_builds_library_json_when_absent(tmp_path):
    """-l pointing to a missing file: statistics are computed from the BAM and the JSON cache is
    written in the reference's schema (utils.py:25-51), then reusable."""
    import json
    lib_json = str(tmp_path / "lib.json")
    out = str(tmp_path / "out.vcf")
    with open(IN_VCF) as inf, open(out, "w") as outf:
        singlesample.sso_genotype(IN_BAM, inf, outf, 20, 1, 1, 1000000, lib_json, False, None, False, 1000, 1e10,
                                  None, 1000, engine=oracle_engine)
    info = json.load(open(lib_json))
    lib = info["NA12878"]["libraryArray"][0]
    assert set(lib) == {"library_name", "readgroups", "read_length", "mean", "sd", "prevalence", "histogram"}
    assert info["NA12878"]["mapped"] == 42801
    out2 = str(tmp_path / "out2.vcf")
    with open(IN_VCF) as inf, open(out2, "w") as outf:
        singlesample.sso_genotype(IN_BAM, inf, outf, 20, 1, 1, 1000000, lib_json, False, None, False, 1000, 1e10,
                                  None, 1000, engine=oracle_engine)
    same_vcf(out, out2)


def test_classic_two_bams_sum_quals(tmp_path):
    """Multi-sample loop + QUAL accumulation (--sum_quals) against the imported reference's output for
    the same call (tests/golden/example.twice.sumquals.gt.vcf.gz)."""
    import gzip
    out = str(tmp_path / "out.vcf")
    with open(IN_VCF) as inf, open(out, "w") as outf:
        classic.sv_genotype(IN_BAM + "," + IN_BAM, inf, outf, 20, 1, 1, 1000000, LIB_JSON, False, None, None, True,
                            None, 1e10, engine=oracle_engine)
    want = gzip.open(os.path.join(HERE, "golden", "example.twice.sumquals.gt.vcf.gz"), "rt").read().split("\n")
    got = [l for l in open(out).read().split("\n") if not l.startswith("##fileDate=")]
    assert len(got) == len(want)
    for i, (x, y) in enumerate(zip(got, want)):
        assert x == y, "line %d\n%s\n%s" % (i + 1, x, y)


@pytest.mark.parametrize("driver", ["classic", "sso"])
def test_native_reader_with_the_oracle_engine(tmp_path, driver, monkeypatch):
    """reader="native" with the geometry predicates in the reader's threads (svt_bam_evidence) hands canonical evidence
    batches to ANY engine: here the CPU oracle, in chunks of 17 units -- the expected VCF, byte for byte, without a GPU."""
    out = str(tmp_path / "out.vcf")
    with open(IN_VCF) as inf, open(out, "w") as outf:
        if driver == "classic":
            monkeypatch.setattr(classic, "CHUNK_UNITS", 17)
            classic.sv_genotype(IN_BAM, inf, outf, 20, 1, 1, 1000000, LIB_JSON, False, None, None, False, None, 1e10,
                                engine=oracle_engine, reader="native")
        else:
            monkeypatch.setattr(singlesample, "CHUNK_UNITS", 17)
            singlesample.sso_genotype(IN_BAM, inf, outf, 20, 1, 1, 1000000, LIB_JSON, False, None, False, 1000, 1e10,
                                      None, 1000, engine=oracle_engine, reader="native")
    same_vcf(EXPECTED, out)
