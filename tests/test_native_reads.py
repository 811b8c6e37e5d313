"""The native C++ BAM reader + fragment summariser (include/svtyper_reads.h) against the Python
implementation (svtyper_amd.bam + fragments + geometry), byte for byte, on the reference's fixture.
CPU only: these entry points need no GPU."""
import json
import os

import numpy as np
import pytest

import goldenio as gio
from svtyper_amd import bam, classic, geometry as geo, hip, library, native_reads as nr, pipeline, singlesample

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


# ------------------------------------------------------------------------------------------
# a synthetic BAM with the record shapes the fixture does not have: records spanning BGZF blocks,
# hard clips, insertions, deletions, N gaps, several SA entries, secondary / supplementary /
# duplicate / unmapped flags, MAPQ 255, B-typed tags in front of the ones that matter
# ------------------------------------------------------------------------------------------
def _synthetic_bam(path, seed=11, n_pairs=700, sample="syn", only_sites=None, sa_first=False, tied_names=False):
    import bamwriter as bw
    rng = np.random.default_rng(seed)
    refs = [("1", 200_000), ("2", 100_000)]
    header = ("@HD\\tVN:1.5\\tSO:coordinate\\n@SQ\\tSN:1\\tLN:200000\\n@SQ\\tSN:2\\tLN:100000\\n"
              "@RG\\tID:rgA\\tSM:%s\\tLB:libA\\n@RG\\tID:rgB\\tSM:%s\\tLB:libB\\n@CO\\tsynthetic\\n" % (sample, sample)).replace("\\t", "\t").replace("\\n", "\n")
    sites = [
        {"id": "d1", "svtype": "DEL", "var_length": 800, "A": {"chrom": "1", "pos": 50_000, "ci": [-5, 5], "is_reverse": False},
         "B": {"chrom": "1", "pos": 50_800, "ci": [-5, 5], "is_reverse": True}},
        {"id": "u1", "svtype": "DUP", "A": {"chrom": "1", "pos": 90_000, "ci": [0, 0], "is_reverse": True},
         "B": {"chrom": "1", "pos": 91_500, "ci": [0, 0], "is_reverse": False}},
        {"id": "i1", "svtype": "INV", "A": {"chrom": "1", "pos": 120_000, "ci": [-10, 10], "is_reverse": False},
         "B": {"chrom": "1", "pos": 123_000, "ci": [-10, 10], "is_reverse": False}},
        {"id": "b1", "svtype": "BND", "A": {"chrom": "1", "pos": 150_000, "ci": [-2, 2], "is_reverse": False},
         "B": {"chrom": "2", "pos": 40_000, "ci": [-2, 2], "is_reverse": True}},
    ]
    for bp in sites:   # the drivers move reverse-strand breakends by one (vcf.get_variant_breakpoints)
        for side in ("A", "B"):
            if bp[side]["is_reverse"]:
                bp[side]["pos"] += 1
    cigars = ["100M", "100M", "100M", "60M40S", "40S60M", "30M2D70M", "50M10I40M", "20H80M", "80M20H", "35M1000N65M",
              "25S50M25S", "10S90M", "70M30S", "45M3D20M5I30M", "15H15S70M"]
    tid_of = {"1": 0, "2": 1}
    recs = []
    covered = list(range(len(sites))) if only_sites is None else list(only_sites)   # the other sites get no read at all
    for k in range(n_pairs):
        bp = sites[covered[k % len(covered)]]
        side = ("A", "B")[int(rng.integers(2))]
        tid = tid_of[bp[side]["chrom"]]
        pos1 = int(bp[side]["pos"] + rng.integers(-450, 150))
        other = "B" if side == "A" else "A"
        far = rng.random() < 0.45                      # mate near the other breakend (discordant) or nearby
        mtid = tid_of[bp[other]["chrom"]] if far else tid
        pos2 = int(bp[other]["pos"] + rng.integers(-150, 450)) if far else pos1 + int(rng.integers(150, 520))
        rev1, rev2 = bool(rng.integers(2)), bool(rng.integers(2))
        mq = [0, 1, 20, 37, 60, 60, 60, 255]
        rg = ("rgA", "rgB")[int(rng.integers(2))]
        name = "q%05d" % int(rng.integers(0, 10 ** 5)) + ("" if rng.random() < 0.9 else "x")
        if tied_names:    # (no extra random draws) names that agree in the seven bytes behind their common prefix
            name = "q%03d000%s" % (int(name[1:6]) % 3, name[1:])
        extra = 0x400 if rng.random() < 0.03 else 0
        for mate, (t, p, r, mt, mp, mr) in enumerate(((tid, pos1, rev1, mtid, pos2, rev2), (mtid, pos2, rev2, tid, pos1, rev1))):
            cigar = cigars[int(rng.integers(len(cigars)))]
            flag = 0x1 | (0x40 if mate == 0 else 0x80) | (0x10 if r else 0) | (0x20 if mr else 0) | extra
            if rng.random() < 0.02:
                flag |= 0x4                             # unmapped but placed
            if rng.random() < 0.02:
                flag |= 0x8
            tags = [("XB", "B", ("s", [1, -2, 3])), ("NM", "C", int(rng.integers(5))), ("RG", "Z", rg)]
            clipped = cigar[-1] in "SH" or cigar.split("M")[0][-1:] in "SH" or "S" in cigar or "H" in cigar
            if clipped and rng.random() < 0.7:          # split alignment near the other breakend
                sa_pos = int(bp[other]["pos"] + rng.integers(-60, 10))
                sa_cig = ["40S60M", "60M40S", "60H40M", "30M70S", "20S60M20S"][int(rng.integers(5))]
                sa = "%s,%d,%s,%s,%d,%d;" % (bp[other]["chrom"], sa_pos + 1, "+-"[int(rng.integers(2))], sa_cig,
                                              mq[int(rng.integers(len(mq)))] % 256, int(rng.integers(4)))
                if rng.random() < 0.2:
                    sa += "2,777,+,50M50S,10,1;"
                if sa_first:      # (no extra random draws: the same reads, SA in front of RG and one more tag behind RG)
                    tags = [tags[0], ("SA", "Z", sa)] + tags[1:] + [("XT", "Z", "tail")]
                else:
                    tags.append(("SA", "Z", sa))
            recs.append(dict(name=name, flag=flag, tid=t, pos=max(0, p), mapq=mq[int(rng.integers(len(mq)))], cigar=cigar,
                             mtid=mt, mpos=max(0, mp), tlen=(mp - p) if t == mt else 0, tags=tags))
            if rng.random() < 0.06:                    # an extra secondary / supplementary record of the same read
                recs.append(dict(recs[-1], flag=flag | (0x100 if rng.random() < 0.5 else 0x800),
                                 pos=max(0, p + int(rng.integers(-30, 30))), cigar=cigars[int(rng.integers(len(cigars)))]))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    bw.write_bam(path, header, refs, recs, block_bytes=int(rng.integers(700, 5000)))
    hist = {str(k): int(1000 * np.exp(-((k - 330) / 70.0) ** 2)) + 1 for k in range(100, 600)}
    lib = lambda nm, rgs: {"library_name": nm, "readgroups": rgs, "read_length": 100, "histogram": hist, "mean": 330.0,
                           "sd": 50.0, "prevalence": 0.5}
    info = {sample: {"mapped": len(recs), "unmapped": 0, "bam": path, "sample_name": sample,
                    "libraryArray": [lib("libA", ["rgA"]), lib("libB", ["rgB"])]}}
    return [{"breakpoint": bp} for bp in sites], info


@pytest.mark.parametrize("seed,mode,max_reads", [(11, nr.COUNT_CLASSIC, None), (12, nr.COUNT_SSO, 1000),
                                                 (13, nr.COUNT_CLASSIC, 90), (14, nr.COUNT_SSO, 200),
                                                 (15, nr.COUNT_CLASSIC, 345), (16, nr.COUNT_SSO, 352)])
def test_synthetic_bam_native_equals_python(tmp_path, seed, mode, max_reads):
    _synthetic_native_equals_python(tmp_path, seed, mode, max_reads, False)


@pytest.mark.parametrize("seed,mode,max_reads", [(11, nr.COUNT_CLASSIC, None), (12, nr.COUNT_SSO, 1000)])
def test_tag_order_does_not_matter(tmp_path, seed, mode, max_reads):
    """the reader looks for RG and SA in ONE walk over a read's tags (SA noted on the way to RG, else searched from behind RG):
    the same reads with SA in front of RG give the same summaries as the Python reader -- and as with SA behind RG"""
    a = _synthetic_native_equals_python(tmp_path, seed, mode, max_reads, True)
    b = _synthetic_native_equals_python(tmp_path, seed, mode, max_reads, False)
    assert a == b


@pytest.mark.parametrize("sa_first", [False, True])
def test_truncated_tag_behind_rg_is_malformed_in_both_tag_orders(tmp_path, sa_first):
    """a tag cut off at the end of the record, behind RG: both readers refuse the read whatever the order of RG and SA (the
    native one walks the tags behind RG even when it met SA on the way to RG)"""
    import bamwriter as bw
    header = "@HD\tVN:1.5\tSO:coordinate\n@SQ\tSN:1\tLN:100000\n@RG\tID:rg\tSM:s\tLB:lib\n"
    sa = ("SA", "Z", "1,52001,+,40S60M,60,0;")
    rg = ("RG", "Z", "rg")
    bad = ("XT", "raw", b"XTZno-terminator")
    good = [dict(name="ok%d" % k, flag=0x1 | 0x40, tid=0, pos=50_000 + k, mapq=60, cigar="100M", mtid=0, mpos=50_300, tlen=400,
                 tags=[rg]) for k in range(3)]
    broken = dict(name="zz", flag=0x1 | 0x40, tid=0, pos=50_010, mapq=60, cigar="60M40S", mtid=0, mpos=50_300, tlen=400,
                  tags=([sa, rg] if sa_first else [rg, sa]) + [bad])
    site = {"id": "d", "svtype": "DEL", "var_length": 800, "A": {"chrom": "1", "pos": 50_050, "ci": [0, 0], "is_reverse": False},
            "B": {"chrom": "1", "pos": 50_851, "ci": [0, 0], "is_reverse": True}}
    hist = {str(k): 10 for k in range(200, 500)}
    info = {"s": {"mapped": 4, "unmapped": 0, "bam": "x", "sample_name": "s", "libraryArray": [
        {"library_name": "lib", "readgroups": ["rg"], "read_length": 100, "histogram": hist, "mean": 350.0, "sd": 50.0, "prevalence": 1.0}]}}
    for name, recs in (("good.bam", good), ("bad.bam", sorted(good + [broken], key=lambda r: r["pos"]))):
        path = str(tmp_path / name)
        bw.write_bam(path, header, [("1", 100000)], recs)
        sample = library.Sample.from_lib_info(bam.AlignmentFile(path), info, 1e-3)
        nbam = nr.NativeBam(path)
        if name == "good.bam":
            assert len(_python_summaries([{"breakpoint": site}], sample, nr.COUNT_SSO, 1000)[1]) == 3
            assert len(_native_summaries([{"breakpoint": site}], sample, nbam, nr.COUNT_SSO, 1000, 2)[1]) == 3
            continue
        with pytest.raises(Exception) as py_err:
            _python_summaries([{"breakpoint": site}], sample, nr.COUNT_SSO, 1000)
        with pytest.raises(hip.SvtyperHipError) as nat_err:
            _native_summaries([{"breakpoint": site}], sample, nbam, nr.COUNT_SSO, 1000, 2)
        assert "malformed" in str(nat_err.value) and not isinstance(py_err.value, hip.SvtyperHipError)


def test_names_that_tie_in_the_sort_key(tmp_path):
    """sorted(query_name) runs on a few bytes behind the unit's common prefix; names that agree there are ordered by the whole
    name afterwards -- the Python reader sorts the names themselves"""
    _synthetic_native_equals_python(tmp_path, 12, nr.COUNT_SSO, 1000, False, tied_names=True)


def _synthetic_native_equals_python(tmp_path, seed, mode, max_reads, sa_first, tied_names=False):
    path = str(tmp_path / ("syn%d%d.bam" % (sa_first, tied_names)))
    sites, info = _synthetic_bam(path, seed, sa_first=sa_first, tied_names=tied_names)
    pybam = bam.AlignmentFile(path)
    sample = library.Sample.from_lib_info(pybam, info, 1e-3)
    nbam = nr.NativeBam(path)
    assert nbam.references == pybam.references and nbam.header["RG"] == pybam.header["RG"]
    want = _python_summaries(sites, sample, mode, max_reads)
    got = _native_summaries(sites, sample, nbam, mode, max_reads, 2)
    assert np.array_equal(got[2], want[2]), "skip flags differ"
    assert np.array_equal(got[0], want[0]), "fragment counts differ: %s vs %s" % (got[0], want[0])
    assert got[1].tobytes() == want[1].tobytes()
    assert len(want[1]) > 100 or want[2].any()
    # the synthetic reads really exercise the split-read path
    assert (want[1]["seq"]["flags"] & 1).any() or (want[1]["clip"]["flags"] & 1).any() or want[2].all()
    return got[0].tobytes(), got[1].tobytes(), got[2].tobytes()


def test_library_statistics_native_equals_python_and_reference(tmp_path):
    """Library.from_bam through the C++ scans == through the Python reader, field for field, including the
    order-sensitive mean / sd."""
    pybam = bam.AlignmentFile(BAM)
    py = library.Sample.from_bam(pybam, 1000000, 1e-3)
    nat = library.Sample.from_bam(pybam, 1000000, 1e-3, native=nr.NativeBam(BAM))
    assert list(py.lib_dict) == list(nat.lib_dict)
    for name in py.lib_dict:
        a, b = py.lib_dict[name], nat.lib_dict[name]
        assert (a.read_length, a.mean, a.sd, a.prevalence, a.readgroups) == (b.read_length, b.mean, b.sd, b.prevalence, b.readgroups)
        assert list(a.hist.items()) == list(b.hist.items())        # same counts in the same (first-seen) order
    # a small num_samp stops the histogram scan early in both readers alike
    for n in (50, 777):
        a = library.Library.from_bam("", pybam, n)
        b = library.Library.from_bam("", pybam, n, native=nr.NativeBam(BAM))
        assert list(a.hist.items()) == list(b.hist.items()) and (a.mean, a.sd) == (b.mean, b.sd)
    # and on the synthetic BAM with two libraries
    path = str(tmp_path / "syn.bam")
    _synthetic_bam(path, seed=31)
    sb = bam.AlignmentFile(path)
    for lib_name in ("libA", "libB"):
        a = library.Library.from_bam(lib_name, sb, 5000)
        b = library.Library.from_bam(lib_name, sb, 5000, native=nr.NativeBam(path))
        assert (a.read_length, a.mean, a.sd, a.prevalence) == (b.read_length, b.mean, b.sd, b.prevalence)
        assert list(a.hist.items()) == list(b.hist.items())
    # (the Python reader's numbers are pinned to the reference's by tests/test_host_pipeline.py)


_DIGEST_CHILD = r'''
import hashlib, os, sys
sys.path.insert(0, os.environ["SVT_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SVT_ROOT"], "tests"))
import test_native_reads as N
from svtyper_amd import bam, library, native_reads as nr
path = sys.argv[1]
sites, info = N._synthetic_bam(path, 21, n_pairs=1500)       # small BGZF blocks: many records straddle them
sample = library.Sample.from_lib_info(bam.AlignmentFile(path), info, 1e-3)
off, frags, skipped = N._native_summaries(sites, sample, nr.NativeBam(path), nr.COUNT_CLASSIC, None, 2)
scan = nr.NativeBam(path).scan_library(["rgA"], 1000000)
print(hashlib.sha256(off.tobytes() + frags.tobytes() + skipped.tobytes()).hexdigest(), repr(scan)[:300])
'''


def test_both_inflate_decoders_give_the_same_summaries(tmp_path):
    """BGZF blocks go through libdeflate when the image has its runtime and through zlib otherwise
    (SVT_INFLATE=zlib forces it); the decoder is chosen once per process, hence the two children."""
    import subprocess
    import sys
    outs = []
    for mode in ("zlib", "default"):
        env = dict(os.environ, SVT_ROOT=os.path.dirname(HERE), SVT_INFLATE=mode)
        r = subprocess.run([sys.executable, "-c", _DIGEST_CHILD, str(tmp_path / ("%s.bam" % mode))], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1] and len(outs[0]) > 64


def _small_bam_with_unplaced_tail(path, n_pairs=40, n_unplaced=30):
    """A coordinate-sorted BAM that ends with unplaced unmapped reads (reference id -1), one of them without an RG
    tag; few enough placed reads that whole-file scans would reach them."""
    import bamwriter as bw
    rng = np.random.default_rng(3)
    header = "@HD\tVN:1.5\tSO:coordinate\n@SQ\tSN:1\tLN:100000\n@RG\tID:rgA\tSM:s\tLB:libA\n"
    recs = []
    for k in range(n_pairs):
        p = 1000 + 50 * k
        ins = int(rng.integers(250, 400))
        for mate, (pos, mpos, rev) in enumerate(((p, p + ins - 100, False), (p + ins - 100, p, True))):
            flag = 0x1 | 0x2 | (0x40 if mate == 0 else 0x80) | (0x10 if rev else 0x20)
            recs.append(dict(name="p%03d" % k, flag=flag, tid=0, pos=pos, mapq=60, cigar="100M", mtid=0, mpos=mpos,
                             tlen=ins if not rev else -ins, tags=[("RG", "Z", "rgA")]))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    for k in range(n_unplaced):                     # after every placed read, as samtools sort leaves them
        tags = [("RG", "Z", "rgA")] if k != 7 else []
        recs.append(dict(name="u%03d" % k, flag=0x1 | 0x4 | 0x8 | (0x40 if k % 2 == 0 else 0x80), tid=-1, pos=-1, mapq=0,
                         cigar="*", mtid=-1, mpos=-1, tlen=0, tags=tags))
    bw.write_bam(path, header, [("1", 100000)], recs, block_bytes=900)
    return 2 * n_pairs


def test_whole_file_scans_stop_at_the_unplaced_reads(tmp_path):
    """pysam's fetch() on an indexed BAM walks reference by reference and never yields the unplaced unmapped reads
    at the end; both readers follow it -- the prevalence denominator, the read length and the -l JSON would differ
    otherwise, and an unplaced read without RG would abort a run the reference completes."""
    path = str(tmp_path / "tail.bam")
    n_placed = _small_bam_with_unplaced_tail(path)
    pybam = bam.AlignmentFile(path)
    assert sum(1 for _ in pybam.fetch()) == n_placed
    assert all(r.reference_id == 0 for r in pybam.fetch())
    for num_samp in (1000000, 5, 0):                 # 0: the reference's `n == num_samp` test never fires -> whole file
        a = library.Library.from_bam("libA", pybam, num_samp)
        b = library.Library.from_bam("libA", pybam, num_samp, native=nr.NativeBam(path))
        assert (a.read_length, a.mean, a.sd, a.prevalence) == (b.read_length, b.mean, b.sd, b.prevalence)
        assert list(a.hist.items()) == list(b.hist.items())
        assert a.prevalence == 1.0 and a.read_length == 100
        assert sum(a.hist.values()) == (5 if num_samp == 5 else n_placed // 2)


def test_odd_records_do_not_end_the_run():
    """An SA-tag MAPQ above 255 is stored as 255 (prob_mapq is exactly 1.0 from 163 on); a mapped primary without a
    CIGAR is simply not a split candidate."""
    import fakereads
    from svtyper_amd import fragments as fr, packer, geometry
    assert packer._mapq(255) == 255 and packer._mapq(300) == 255 and geometry._mapq8(100000) == 255
    assert 1 - 10 ** (-163 / 10.0) == 1.0 and 1 - 10 ** (-255 / 10.0) == 1.0 and 1 - 10 ** (-1000 / 10.0) == 1.0
    with pytest.raises(ValueError):
        packer._mapq(-1)

    class _L:
        mean, sd = 300.0, 50.0
    r = fakereads.FakeRead("q", 0x1 | 0x40, "1", 1000, "", 60)
    assert not r.cigar
    f = fr.SamFragment(r, _L())
    assert f.num_primary == 1 and f.split_reads == []


def _python_records(sites, sample, mode, max_reads, min_aligned=20, split_slop=3):
    """the units' evidence records through the Python reader + packer (the reference's predicates restated in packer.py)"""
    from svtyper_amd import evidence as ev, packer
    lib_index = {id(lib): i for i, lib in enumerate(sample.lib_dict.values())}
    offs, recs, skipped = [0], [], []
    for s in sites:
        bp = s["breakpoint"]
        frags, many = (classic.gather_all_reads if mode == nr.COUNT_CLASSIC else singlesample.gather_reads)(sample, bp, max_reads)
        a = packer.pack_fragments(frags, bp, lib_index, min_aligned, split_slop) if frags and not many else np.zeros(0, ev.RECORD_DTYPE)
        recs.append(a)
        offs.append(offs[-1] + len(a))
        skipped.append(1 if many else 0)
    return np.asarray(offs, np.uint64), np.concatenate(recs), np.asarray(skipped, np.uint8)


def _native_records(sites, sample, nbam, mode, max_reads, threads, min_aligned=20, split_slop=3):
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
    flank = [float(lib.mean) + float(lib.sd) * 3 for lib in libs]
    return nbam.evidence(win, bps, rgs, rg_lib, max_reads, mode, flank, min_aligned, split_slop, threads)


@pytest.mark.parametrize("mode,max_reads,threads,min_aligned", [
    (nr.COUNT_CLASSIC, None, 3, 20), (nr.COUNT_SSO, 1000, 2, 20), (nr.COUNT_SSO, 120, 1, 20), (nr.COUNT_CLASSIC, None, 2, 34)])
def test_evidence_records_equal_the_python_packer(setup, mode, max_reads, threads, min_aligned):
    """svt_bam_evidence (the geometry predicates of svt_geometry_math.h in the reader's threads) against the Python reader +
    packer.py on the fixture's 211 sites: the 16-byte records, their offsets and the skip flags, byte for byte."""
    sites, sample, nbam = setup
    want = _python_records(sites, sample, mode, max_reads, min_aligned)
    got = _native_records(sites, sample, nbam, mode, max_reads, threads, min_aligned)
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0])
    assert got[1].tobytes() == want[1].tobytes()
    assert len(want[1]) > 5000 or want[2].any()
    assert (want[1]["flags"] & 1).any() and want[1]["seq_l"].any() and want[1]["rs_a"].any()      # (the predicates are exercised)


def test_evidence_rejects_a_library_outside_the_table(setup):
    sites, sample, nbam = setup
    from svtyper_amd import hip
    tid_of = nbam.gettid
    bps = np.concatenate([geo.breakpoint_record(s["breakpoint"], tid_of) for s in sites[:3]])
    win = np.zeros(3, nr.FETCH_DTYPE)
    for k, s in enumerate(sites[:3]):
        bp = s["breakpoint"]
        win["tid_a"][k] = win["tid_b"][k] = tid_of(bp["A"]["chrom"])
        win["lo_a"][k], win["hi_a"][k] = max(0, bp["A"]["pos"] - 500), bp["A"]["pos"] + 500
        win["lo_b"][k], win["hi_b"][k] = max(0, bp["B"]["pos"] - 500), bp["B"]["pos"] + 500
    rgs = list(sample.rg_to_lib.keys())
    with pytest.raises(hip.SvtyperHipError):
        nbam.evidence(win, bps, rgs, [5] * len(rgs), None, nr.COUNT_CLASSIC, [400.0], 20, 3, 1)     # library 5 of a table of 1
