"""The native C++ BAM reader + fragment summariser (include/svtyper_reads.h) against the Python
implementation (svtyper_amd.bam + fragments + geometry), byte for byte, on the reference's fixture.
CPU only: these entry points need no GPU."""
import json
import os

import numpy as np
import pytest

import goldenio as gio
from svtyper_amd import bam, classic, geometry as geo, library, native_reads as nr, pipeline, singlesample

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")
BAM = os.path.join(DATA, "NA12878.target_loci.sorted.bam")


@pytest.fixture(scope="module")
def setup():
    g = gio.load("fixture_sites.json.gz")
    info = json.load(open(os.path.join(DATA, "NA12878.bam.json")))
    pybam = bam.AlignmentFile(BAM)
    sample = library.Sample.from_lib_info(pybam, info, 1e-3)
    nbam = nr.NativeBam(BAM)
    return g["sites"], sample, nbam


def test_header_matches_python_reader(setup):
    _, sample, nbam = setup
    assert nbam.references == sample.bam.references and nbam.lengths == sample.bam.lengths
    assert nbam.header["RG"] == sample.bam.header["RG"]
    assert nbam.gettid("2") == sample.bam.gettid("2") and nbam.gettid("nope") == -1


def _python_summaries(sites, sample, mode, max_reads):
    lib_index = {id(lib): i for i, lib in enumerate(sample.lib_dict.values())}
    tid_of = sample.bam.gettid
    offs, frs, skipped = [0], [], []
    for s in sites:
        bp = s["breakpoint"]
        if mode == nr.COUNT_CLASSIC:
            frags, many = classic.gather_all_reads(sample, bp, max_reads)
        else:
            frags, many = singlesample.gather_reads(sample, bp, max_reads)
        a = geo.summarise_fragments(frags, bp, lib_index, tid_of) if not many else np.zeros(0, geo.FRAGMENT_DTYPE)
        frs.append(a)
        offs.append(offs[-1] + len(a))
        skipped.append(1 if many else 0)
    return np.asarray(offs, np.uint64), np.concatenate(frs), np.asarray(skipped, np.uint8)


def _native_summaries(sites, sample, nbam, mode, max_reads, threads):
    tid_of = nbam.gettid
    bps = np.concatenate([geo.breakpoint_record(s["breakpoint"], tid_of) for s in sites])
    win = np.zeros(len(sites), nr.FETCH_DTYPE)
    for k, s in enumerate(sites):
        bp = s["breakpoint"]
        for side, (t, lo, hi) in (("A", ("tid_a", "lo_a", "hi_a")), ("B", ("tid_b", "lo_b", "hi_b"))):
            chrom, a, b = pipeline.fetch_window(sample, bp[side]["chrom"], bp[side]["pos"], bp[side]["ci"],
                                                as_int=(mode == nr.COUNT_SSO))
            win[t][k], win[lo][k], win[hi][k] = tid_of(chrom), int(a), int(b)
    rgs = list(sample.rg_to_lib.keys())
    libs = list(sample.lib_dict.values())
    rg_lib = [libs.index(sample.rg_to_lib[rg]) if sample.rg_to_lib[rg].name in sample.active_libs else -1 for rg in rgs]
    return nbam.summarise(win, bps, rgs, rg_lib, max_reads, mode, threads)


@pytest.mark.parametrize("mode,max_reads,threads", [
    (nr.COUNT_CLASSIC, None, 1), (nr.COUNT_CLASSIC, None, 4), (nr.COUNT_CLASSIC, 150, 3),
    (nr.COUNT_SSO, 1000, 2), (nr.COUNT_SSO, 120, 1)])
def test_summaries_equal_python(setup, mode, max_reads, threads):
    sites, sample, nbam = setup
    want = _python_summaries(sites, sample, mode, max_reads)
    got = _native_summaries(sites, sample, nbam, mode, max_reads, threads)
    assert np.array_equal(got[2], want[2]), "skip flags differ"
    assert np.array_equal(got[0], want[0]), "fragment counts differ"
    assert got[1].tobytes() == want[1].tobytes()
    if max_reads is not None and max_reads < 1000:
        assert want[2].any() and not want[2].all()   # the threshold really splits the sites
    assert len(want[1]) > 5000 or want[2].any()
