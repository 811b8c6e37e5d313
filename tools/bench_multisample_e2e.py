#!/usr/bin/env python
"""End-to-end classic (multi-sample) run on synthetic BAMs: S samples x V variants through sv_genotype with the
native reader; reports where the wall time goes (C++ summariser + device vs the Python VCF layer)."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_native_reads as N
import test_hip_geometry as G
from svtyper_amd import classic, native_reads, pipeline

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
tmp = tempfile.mkdtemp()
paths, info = [], {}
for k in range(S):
    p = os.path.join(tmp, "s%02d.bam" % k)
    _, inf = N._synthetic_bam(p, seed=100 + k, n_pairs=900, sample="smp%02d" % k)
    info.update(inf)
    paths.append(p)
lib_json = os.path.join(tmp, "libs.json")
json.dump(info, open(lib_json, "w"))
import pathlib
_, vcf_path, _ = G._synthetic_case(pathlib.Path(tmp))
lines = open(vcf_path).read().splitlines(True)
hdr = [l for l in lines if l.startswith("#")]; body = [l for l in lines if not l.startswith("#")]
big = os.path.join(tmp, "big.vcf"); open(big, "w").write("".join(hdr) + "".join(body * REP))
t_c = [0.0]
orig_sum = native_reads.NativeBam.summarise
def timed_sum(self, *a, **k):
    t0 = time.perf_counter(); r = orig_sum(self, *a, **k); t_c[0] += time.perf_counter() - t0; return r
native_reads.NativeBam.summarise = timed_sum
t_d = [0.0]
eng = pipeline.default_engine()
orig_gf = eng.genotype_fragments
def timed_gf(*a, **k):
    t0 = time.perf_counter(); r = orig_gf(*a, **k); t_d[0] += time.perf_counter() - t0; return r
eng.genotype_fragments = timed_gf
for rep in range(2):
    t_c[0] = t_d[0] = 0.0
    t0 = time.perf_counter()
    with open(big) as inf, open(os.path.join(tmp, "out.vcf"), "w") as outf:
        classic.sv_genotype(",".join(paths), inf, outf, 20, 1, 1, 1000000, lib_json, False, None, None, False, None, 1e10,
                            engine=eng, geometry="device", reader="native")
    dt = time.perf_counter() - t0
    nv = len(body) * REP
    print("%d samples x %d variants = %d units: %.2f s wall | C++ summarise %.2f s, device %.2f s (both on the worker thread) "
          "| %.0f sites/s, %.0f units/s" % (S, nv, nv * S, dt, t_c[0], t_d[0], nv / dt, nv * S / dt))
