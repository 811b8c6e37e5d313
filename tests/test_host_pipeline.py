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


def run_classic(out, engine):
    with open(IN_VCF) as inf, open(out, "w") as outf:
        classic.sv_genotype(bam_string=IN_BAM, vcf_in=inf, vcf_out=outf, min_aligned=20, split_weight=1,
                            disc_weight=1, num_samp=1000000, lib_info_path=LIB_JSON, debug=False,
                            alignment_outpath=None, ref_fasta=None, sum_quals=False, max_reads=None,
                            max_ci_dist=1e10, engine=engine)


def run_sso(out, engine, cores):
    with open(IN_VCF) as inf, open(out, "w") as outf:
        singlesample.sso_genotype(bam_string=IN_BAM, vcf_in=inf, vcf_out=outf, min_aligned=20, split_weight=1,
                                  disc_weight=1, num_samp=1000000, lib_info_path=LIB_JSON, debug=False,
                                  ref_fasta=None, sum_quals=False, max_reads=1000, max_ci_dist=1e10, cores=cores,
                                  batch_size=1000, engine=engine)


def test_classic_integration_oracle_engine(tmp_path):
    out = str(tmp_path / "out.vcf")
    run_classic(out, oracle_engine)
    same_vcf(EXPECTED, out)


@pytest.mark.parametrize("cores", [None, 1])
def test_sso_integration_oracle_engine(tmp_path, cores):
    out = str(tmp_path / "out.vcf")
    run_sso(out, oracle_engine, cores)
    same_vcf(EXPECTED, out)


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
