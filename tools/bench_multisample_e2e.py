#!/usr/bin/env python
"""End-to-end classic (multi-sample) run on synthetic BAMs: S samples x V variants through sv_genotype with the native reader,
three ways: the readers' record arrays handed over as they are (sample-major, svt_batch_create_segments), interleaved site-major
on the host first, and as 128-byte summaries with the geometry on the device.  GPU box only."""
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
eng = pipeline.default_engine()


class Interleaving:
    """the same engine without `accepts_sample_major`: NativeUnitCollector interleaves the samples' records site-major on the host
    (what every joint run did before svt_batch_create_segments)"""
    supports_site_qual = True

    def __call__(self, batch, flags=0, site_qual=None):
        return eng(batch, flags, site_qual=site_qual)


def run(engine, geometry):
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        with open(big) as inf, open(os.path.join(tmp, "out_%s.vcf" % geometry), "w") as outf:
            classic.sv_genotype(",".join(paths), inf, outf, 20, 1, 1, 1000000, lib_json, False, None, None, False, None, 1e10,
                                engine=engine, geometry=geometry, reader="native")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, open(os.path.join(tmp, "out_%s.vcf" % geometry)).read()


nv = len(body) * REP
rows = []
for name, engine, geometry in (("records in segments, sample-major (svt_batch_create_segments)", eng, "host"),
                               ("records interleaved site-major on the host", Interleaving(), "host"),
                               ("128-byte summaries, geometry on the device, one batch per sample", eng, "device")):
    dt, text = run(engine, geometry)
    rows.append(text)
    print("%d samples x %d variants = %d units | %-72s %.3f s wall = %.0f sites/s, %.0f units/s" % (S, nv, nv * S, name, dt, nv / dt, nv * S / dt), flush=True)
strip = lambda t: [l for l in t.split("\n") if not l.startswith("##fileDate=")]
print("VCFs identical:", strip(rows[0]) == strip(rows[1]) == strip(rows[2]))
os.environ["SVT_TRACE"] = "1"
print("--- stage laps of one more run of the first form (stderr) ---", flush=True)
run(eng, "host")
