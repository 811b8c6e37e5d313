#!/usr/bin/env python
"""End-to-end wall time of sso_genotype on the reference's fixture (211 breakpoints) for the three host
configurations; all three must reproduce example.gt.vcf."""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_host_pipeline as T
from svtyper_amd import pipeline
engine = pipeline.default_engine()
for label, kw in (("python reader, host geometry", dict()), ("python reader, device geometry", dict(geometry="device")),
                  ("native reader, device geometry", dict(geometry="device", reader="native"))):
    best = 1e9
    for rep in range(3):
        out = "/tmp/e2e_out.vcf"
        t0 = time.perf_counter()
        with open(T.IN_VCF) as inf, open(out, "w") as outf, open(os.devnull, "w") as null:
            old = sys.stderr; sys.stderr = null
            try:
                T.singlesample.sso_genotype(T.IN_BAM, inf, outf, 20, 1, 1, 1000000, T.LIB_JSON, False, None, False, 1000,
                                            1e10, None, 1000, engine=engine, **kw)
            finally:
                sys.stderr = old
        best = min(best, time.perf_counter() - t0)
        T.same_vcf(T.EXPECTED, out)
    print("%-34s %.3f s for 211 breakpoints = %.0f breakpoints/s" % (label, best, 211 / best))
