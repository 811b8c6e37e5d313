#!/usr/bin/env python
"""The two public drivers end to end with stage laps (SVT_TRACE=1 / SVT_TRACE_VCF=1 on stderr): bench.py's `driver_sso` and
`driver_classic_8bam` workloads, one traced run each after the timed ones.  GPU box only (the default engine)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if "--r05-workload" in sys.argv:
    # the workload profiles/r05_multisample_e2e.txt was measured on (146 k units/s through round 5's per-line driver): 8 BAMs of
    # tests/test_native_reads.py::_synthetic_bam (900 pairs around four sites, two libraries each), the five-line VCF of
    # tests/test_library_groups.py x 2000 = 10 000 variant lines (8 000 sites: a BND pair is two lines) x 8 samples
    import io
    import tempfile
    import time
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_library_groups as G
    from svtyper_amd import classic
    with tempfile.TemporaryDirectory() as tmp:
        import pathlib
        paths, libs = G.cohort(pathlib.Path(tmp), 8, 900)
        text = G.vcf_text(2000)
        for bulk in ("1", "0"):
            os.environ["SVT_BULK_VCF"] = bulk
            best = None
            for _ in range(3):
                out = G.Sink()
                t0 = time.perf_counter()
                classic.sv_genotype(",".join(paths), io.StringIO(text), out, 20, 1, 1, 1000000, libs, False, None, None, False, None, 1e10)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            print("8 samples x 10000 variant lines (8000 sites, 64000 units) | %s route: %.3f s wall = %.0f lines/s, %.0f units/s (r05 counted lines x samples: %.0f)"
                  % ("bulk" if bulk == "1" else "per-line", best, 10000 / best, 64000 / best, 80000 / best), flush=True)
    sys.exit(0)

legs = bench.driver_legs()
print(json.dumps(legs, indent=1))
if "--trace" in sys.argv:
    os.environ["SVT_TRACE"] = os.environ["SVT_TRACE_VCF"] = "1"
    import io
    import tempfile
    from svtyper_amd import classic, singlesample
    data = os.path.join(ROOT, "tests", "data")
    lines = open(os.path.join(data, "example.vcf")).readlines()
    head = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if not l.startswith("#")]

    class Sink(io.StringIO):
        def close(self):
            pass
    import time
    sys.stderr.write("---- sso, fixture x 100 ----\n")
    t0 = time.perf_counter()
    singlesample.sso_genotype(os.path.join(data, "NA12878.target_loci.sorted.bam"), io.StringIO("".join(head) + "".join(body * 100)), Sink(),
                              20, 1, 1, 1000000, os.path.join(data, "NA12878.bam.json"), False, None, False, 1000, 1e10, None, 1000)
    sys.stderr.write("wall %.1f ms\n" % ((time.perf_counter() - t0) * 1e3))
    with tempfile.TemporaryDirectory() as tmp:
        paths, info, sites = [], {}, None
        for k in range(8):
            path = os.path.join(tmp, "s%d.bam" % k)
            inf, sites, _ = bench._wgs_like_bam(path, genome=300_000, seed=40 + k, sample="smp%d" % k)
            info.update(inf)
            paths.append(path)
        libs = os.path.join(tmp, "libs.json")
        json.dump(info, open(libs, "w"))
        vlines = ["1\t%d\t%s\tN\t<DEL>\t0\t.\tSVTYPE=DEL;SVLEN=-%d;END=%d;STR=+-:8;CIPOS=-10,10;CIEND=-10,10;SU=8;PE=6;SR=2\n"
                  % (bp["A"]["pos"], bp["id"], bp["var_length"], bp["A"]["pos"] + bp["var_length"]) for bp in sites]
        reps = -(-10_500 // len(vlines))
        vtext = "".join(l for l in head if l.startswith("##")) + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" + "".join(vlines * reps)
        for rep in range(2):
            sys.stderr.write("---- classic, 8 BAMs x %d lines (run %d) ----\n" % (len(vlines) * reps, rep))
            t0 = time.perf_counter()
            classic.sv_genotype(",".join(paths), io.StringIO(vtext), Sink(), 20, 1, 1, 1000000, libs, False, None, None, False, None, 1e10)
            sys.stderr.write("wall %.1f ms\n" % ((time.perf_counter() - t0) * 1e3))
