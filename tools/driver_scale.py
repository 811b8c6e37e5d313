#!/usr/bin/env python
"""The singlesample driver end to end on a VCF of many sites (the fixture's 211 breakpoints x N): VCF text in -> VCF text out
through reader="native", geometry="device".  Prints the wall time, sites/s and the top of a cProfile of the same run: what a
user of `svtyper-sso` waits for once the reader and the device stages are fast.  GPU box only (the HIP engine)."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svtyper_amd import singlesample  # noqa: E402

REPEAT = int(sys.argv[1]) if len(sys.argv) > 1 else 100
data = os.path.join(ROOT, "tests", "data")
with open(os.path.join(data, "example.vcf")) as f:
    lines = f.readlines()
head = [l for l in lines if l.startswith("#")]
body = [l for l in lines if not l.startswith("#")]
text = "".join(head) + "".join(body * REPEAT)
bam = os.path.join(data, "NA12878.target_loci.sorted.bam")
lib = os.path.join(data, "NA12878.bam.json")


def run():
    out = io.StringIO()
    with open(os.devnull, "w") as null:
        old, sys.stderr = sys.stderr, null
        try:
            singlesample.sso_genotype(bam, io.StringIO(text), out, 20, 1, 1, 1000000, lib, False, None, False, 1000, 1e10, None, 1000,
                                      geometry="device", reader="native")
        finally:
            sys.stderr = old
    return out.getvalue()


run()                                       # (library load, device context, pools)
best = None
for _ in range(3):
    t0 = time.perf_counter()
    vcf = run()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
n = len(body) * REPEAT
print("%d variant lines: %.1f ms = %.0f sites/s (%d bytes of VCF out)" % (n, best * 1e3, n / best, len(vcf)))
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue())
