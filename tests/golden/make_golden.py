#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (dev container only).

    python tests/golden/make_golden.py

Outputs (committed; the reference itself never travels):
  bayes_grid.json.gz     (tallies, svtype, weights) -> reference bayesian_genotype() result dicts
                         + bayes_gt() log-likelihoods as hex floats            [statistics.py, singlesample.py:406-473]
  fixture_sites.json.gz  the 211 breakpoints of the reference's own fixture (tests/data): breakpoint
                         dict, packed evidence records (from the reference's fragment objects and
                         predicates), reference tallies in both associations, reference result
  library_from_bam.json.gz  Sample.from_bam() statistics of the fixture BAM
  fake_sites.json.gz     synthetic fake-read sites: libraries, breakpoint, the reads themselves,
                         packed records, reference tallies + result

Floats are stored as float.hex() strings so they round-trip bit-exactly.
"""
from __future__ import annotations

import contextlib
import gzip
import io
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import refload  # noqa: E402
import fakereads  # noqa: E402
from svtyper_amd import bam as bam_module  # noqa: E402
from svtyper_amd import packer  # noqa: E402

DATA = os.path.join(ROOT, "tests", "data")
BAM = os.path.join(DATA, "NA12878.target_loci.sorted.bam")
VCF = os.path.join(DATA, "example.vcf")
LIBJSON = os.path.join(DATA, "NA12878.bam.json")

TALLIES = ("ref_seq", "alt_seq", "alt_clip", "ref_span", "alt_span")


def hx(x):
    return float(x).hex()


def dump(name, obj):
    path = os.path.join(HERE, name)
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(obj, separators=(",", ":"), sort_keys=True).encode())
    print("wrote %s (%d bytes)" % (name, os.path.getsize(path)))


def result_to_json(result):
    fm = dict(result["formats"])
    out = {"qual": hx(result["qual"]) if isinstance(result["qual"], float) else result["qual"], "formats": {}}
    for k, v in fm.items():
        out["formats"][k] = {"f": hx(v)} if isinstance(v, float) else v
    return out


def blank_like(ref):
    return result_to_json(ref.singlesample.blank_genotype_result())


# ------------------------------------------------------------------------------------------
def make_bayes_grid(ref):
    rng = random.Random(20260927)
    cases = []

    def add(t, svtype, sw, dw):
        counts = dict(zip(TALLIES, t))
        bp = {"id": "g", "svtype": svtype}
        total = sum(counts.values())
        if total == 0:
            return
        res = ref.singlesample.bayesian_genotype(bp, counts, sw, dw, False)
        a = counts["alt_seq"] + counts["alt_clip"]
        QR = int(sw * counts["ref_seq"]) + int(dw * counts["ref_span"])
        QA = int(sw * a) + int(dw * counts["alt_span"])
        gl = ref.statistics.bayes_gt(QR, QA, svtype == "DUP")
        cases.append({"tallies": [hx(x) for x in t], "svtype": svtype, "sw": hx(sw), "dw": hx(dw),
                      "gl": [hx(x) for x in gl], "result": result_to_json(res)})

    grid = [0, 1, 2, 3, 4, 5, 7, 10, 15, 20, 33, 50, 77, 100, 150, 174, 200, 300]
    for svtype in ("DEL", "DUP", "INV", "BND"):
        for qr in grid:
            for qa in grid:
                add([float(qr), float(qa), 0.0, 0.0, 0.0], svtype, 1, 1)
    # underflow of sum(10**GL): deep duplications and the boundary around it
    for qa in (600, 640, 650, 660, 670, 675, 676, 677, 678, 679, 680, 700, 900):
        add([0.0, 0.0, 0.0, 0.0, float(qa)], "DUP", 1, 1)
        add([1.0, 0.0, 0.0, 0.0, float(qa)], "DUP", 1, 1)
    for qr in (3000, 6000, 7000, 7500):
        add([float(qr), 0.0, 0.0, 0.0, 3.0], "DEL", 1, 1)
    # realistic fractional tallies, float weights
    for _ in range(1500):
        t = [rng.uniform(0, 60) * (rng.random() < 0.8) for _ in range(5)]
        t = [x * rng.choice([0.999999, 0.99, 0.9, 1.0]) for x in t]
        sw, dw = rng.choice([(1, 1), (1, 1), (0.5, 1.0), (1.5, 0.7), (2, 3)])
        add(t, rng.choice(["DEL", "DUP", "INV", "BND"]), sw, dw)
    dump("bayes_grid.json.gz", {"cases": cases})


# ------------------------------------------------------------------------------------------
def lib_tables(ref_libs):
    """reference Library objects -> ([json], {id(lib): index})"""
    out, index = [], {}
    for i, lib in enumerate(ref_libs):
        index[id(lib)] = i
        out.append({"name": lib.name, "readgroups": list(lib.readgroups), "mean": hx(lib.mean), "sd": hx(lib.sd),
                    "read_length": lib.read_length, "hist": {str(k): int(v) for k, v in lib.hist.items()}})
    return out, index


def make_fixture(ref):
    ss = ref.singlesample
    sample = ss.setup_sample(BAM, LIBJSON, None, 1000000, 20)
    libs = list(sample.lib_dict.values())
    libs_json, lib_index = lib_tables(libs)
    vcf = ss.init_vcf(VCF, sample, "/nonexistent-scratch")

    # classic association, pre-zeroing, from the reference's own --debug prints
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), open(VCF) as fin, open(os.devnull, "w") as fout:
        ref.classic.sv_genotype(BAM, fin, fout, 20, 1, 1, 1000000, LIBJSON, True, None, None, False, None, 1e10)
    classic_raw = []
    cur = {}
    for line in buf.getvalue().splitlines():
        for key in ("ref_span", "alt_span", "ref_seq", "alt_seq", "alt_clip"):
            if line.startswith(key + ":"):
                cur[key] = float(line.split(":", 1)[1])
        if len(cur) == 5:
            classic_raw.append(cur)
            cur = {}

    sites = []
    bnd_cache = {}
    for vline in ref.utils.vcf_variants(VCF):
        v = vline.rstrip().split("\t")
        variant = ref.parsers.Variant(v, vcf)
        if not variant.has_svtype() or not variant.is_valid_svtype():
            continue
        bp = vcf.get_variant_breakpoints(variant, 1e10)
        if variant.get_svtype() == "BND":
            if variant.info["MATEID"] in bnd_cache:
                variant = bnd_cache[variant.info["MATEID"]]
            else:
                bnd_cache[variant.var_id] = variant
                continue
        if bp is None:
            continue
        regions = ss.get_breakpoint_regions(bp, sample, 3)
        frags, many = ss.gather_reads(sample.bam, bp["id"], regions, sample.rg_to_lib, sample.active_libs, 1000)
        assert not many
        recs = packer.pack_fragments(frags, bp, lib_index, 20, 3)
        counts = ss.tally_variant_read_fragments(3, 20, bp, frags, False)
        if sum(counts.values()) == 0:
            result = blank_like(ref)
        else:
            result = result_to_json(ss.bayesian_genotype(bp, counts, 1, 1, False))
        k = len(sites)
        sites.append({
            "breakpoint": bp,
            "n_fragments": len(frags),
            "records": [[int(x) for x in row] for row in recs.tolist()],
            "tallies_sso": {t: hx(counts[t]) for t in TALLIES},                # after the zeroing rules
            "tallies_classic_raw": {t: hx(classic_raw[k][t]) for t in TALLIES},  # before the zeroing rules
            "result": result,
        })
    assert len(sites) == len(classic_raw) == 211, (len(sites), len(classic_raw))
    dump("fixture_sites.json.gz", {"libraries": libs_json, "sites": sites, "min_aligned": 20, "split_slop": 3})


# ------------------------------------------------------------------------------------------
def make_fake(ref, n_sites=420):
    rng = random.Random(4242)
    groups = []
    for g in range(6):
        libs = fakereads.make_libraries(rng, rng.choice([1, 1, 2, 3]))
        ref_libs = [ref.parsers.Library(name, None, rgs, rl, dict(hist), None, mean, sd, 1.0, 0)
                    for (name, rgs, mean, sd, rl, hist) in libs]
        rg_to_lib = {rg: L for L, spec in zip(ref_libs, libs) for rg in spec[1]}
        libs_json, lib_index = lib_tables(ref_libs)
        sites = []
        for s in range(n_sites // 6):
            bp, reads = fakereads.make_site(rng, "s%d_%d" % (g, s), libs)
            frags = {}
            for r in reads:                      # as gather_reads does (singlesample.py:194-203)
                lib = rg_to_lib[r.get_tag("RG")]
                if r.query_name in frags:
                    frags[r.query_name].add_read(r)
                else:
                    frags[r.query_name] = ref.parsers.SamFragment(r, lib)
            recs = packer.pack_fragments(frags, bp, lib_index, 20, 3)
            counts = ref.singlesample.tally_variant_read_fragments(3, 20, bp, frags, False)
            if sum(counts.values()) == 0:
                result = blank_like(ref)
            else:
                result = result_to_json(ref.singlesample.bayesian_genotype(bp, counts, 1, 1, False))
            sites.append({
                "breakpoint": bp,
                "reads": [list(r.astuple()) for r in reads],
                "records": [[int(x) for x in row] for row in recs.tolist()],
                "tallies_sso": {t: hx(counts[t]) for t in TALLIES},
                "result": result,
            })
        groups.append({"libraries": libs_json, "sites": sites})
    dump("fake_sites.json.gz", {"groups": groups, "read_fields": list(fakereads.READ_FIELDS),
                                "min_aligned": 20, "split_slop": 3})


def make_library_from_bam(ref):
    """Library statistics built empirically from the fixture BAM by the reference
    (parsers.py:472-583, statistics.py:40-121) -- pins svtyper_amd.library.Library.from_bam."""
    bam = bam_module.AlignmentFile(BAM)
    sample = ref.parsers.Sample.from_bam(bam, 1000000, 1e-3)
    libs = []
    for lib in sample.lib_dict.values():
        libs.append({"name": lib.name, "readgroups": list(lib.readgroups), "read_length": lib.read_length,
                     "mean": hx(lib.mean), "sd": hx(lib.sd), "prevalence": hx(lib.prevalence),
                     "hist": {str(k): int(v) for k, v in lib.hist.items()}})
    dump("library_from_bam.json.gz", {"sample": sample.name, "active_libs": list(sample.active_libs),
                                      "fetch_flank_z3": hx(sample.get_fetch_flank(3)), "libraries": libs,
                                      "mapped": sample.bam_mapped, "unmapped": sample.bam_unmapped})


def make_multisample_vcf(ref):
    """classic.sv_genotype with the fixture BAM given twice (two 'samples' sharing one column) and
    --sum_quals: pins the per-sample loop, QUAL accumulation and the blank-result QUAL reset
    (classic.py:216-217,279,485,498)."""
    out = io.StringIO()
    out.close = lambda: None
    with open(VCF) as fin:
        ref.classic.sv_genotype(BAM + "," + BAM, fin, out, 20, 1, 1, 1000000, LIBJSON, False, None, None, True,
                                None, 1e10)
    text = "\n".join(l for l in out.getvalue().split("\n") if not l.startswith("##fileDate="))
    path = os.path.join(HERE, "example.twice.sumquals.gt.vcf.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(text.encode())
    print("wrote %s (%d bytes)" % (os.path.basename(path), os.path.getsize(path)))


def make_three_sample_vcf(ref):
    """classic.sv_genotype over three BAMs with a blank sample in the middle, with and without --sum_quals: pins the
    running QUAL over a site's samples and its reset by a sample without evidence (classic.py:216-217,485,498)."""
    import tempfile
    from test_multisample_qual import three_sample_case
    with tempfile.TemporaryDirectory() as wd:
        bams, vcf_path, lib_json = three_sample_case(wd)
        for sum_quals, name in ((True, "three.sumquals.gt.vcf.gz"), (False, "three.gt.vcf.gz")):
            out = io.StringIO()
            out.close = lambda: None
            with open(vcf_path) as fin:
                ref.classic.sv_genotype(bams, fin, out, 20, 1, 1, 1000000, lib_json, False, None, None, sum_quals, None, 1e10)
            text = "\n".join(l for l in out.getvalue().split("\n") if not l.startswith("##fileDate="))
            path = os.path.join(HERE, name)
            with gzip.GzipFile(path, "wb", mtime=0) as f:
                f.write(text.encode())
            print("wrote %s (%d bytes)" % (name, os.path.getsize(path)))


if __name__ == "__main__":
    ref = refload.load_reference(pysam_module=bam_module)
    make_bayes_grid(ref)
    make_fixture(ref)
    make_fake(ref)
    make_library_from_bam(ref)
    make_multisample_vcf(ref)
    make_three_sample_vcf(ref)
