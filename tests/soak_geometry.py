#!/usr/bin/env python
"""Open-ended check of the stages in front of the likelihood path: random synthetic BAMs (tests/bamwriter.py) through
sso_genotype / sv_genotype with (Python reader, host geometry) and with (native reader, device geometry); the two VCFs
must be identical.  Usage: tests/soak_geometry.py [seconds]"""
import io, json, os, pathlib, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_native_reads as N
import test_hip_geometry as G
import test_host_pipeline as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # seeds derive from the iteration number
t0, it = time.time(), first
devnull = open(os.devnull, "w")
while time.time() - t0 < budget:
    tmp = pathlib.Path(tempfile.mkdtemp())
    bam_path = str(tmp / "syn.bam")
    _, info = N._synthetic_bam(bam_path, seed=5000 + it, n_pairs=300 + 97 * (it % 9))
    lib_json = str(tmp / "syn.json"); json.dump(info, open(lib_json, "w"))
    _, vcf_path, _ = G._synthetic_case(tmp)            # writes its own BAM too; only the VCF is used
    outs = []
    for kw in ({}, dict(geometry="device", reader="native")):
        for driver in ("sso", "classic"):
            out = str(tmp / ("%s_%s.vcf" % (driver, kw.get("reader", "python"))))
            old = sys.stderr; sys.stderr = devnull
            try:
                with open(vcf_path) as inf, open(out, "w") as outf:
                    if driver == "classic":
                        T.classic.sv_genotype(bam_path, inf, outf, 20, 1, 1, 1000000, lib_json, False, None, None, False,
                                              None if it % 2 else 400, 1e10, **kw)
                    else:
                        T.singlesample.sso_genotype(bam_path, inf, outf, 20, 1, 1, 1000000, lib_json, False, None, False,
                                                    1000 if it % 2 else 500, 1e10, None, 1000, **kw)
            finally:
                sys.stderr = old
            outs.append(open(out).read())
    if outs[0] != outs[2] or outs[1] != outs[3]:
        print("MISMATCH at iteration %d (files under %s)" % (it, tmp))
        sys.exit(1)
    it += 1
print("geometry soak ok: iterations %d..%d (synthetic BAMs) x 2 drivers, %.0f s" % (first, it - 1, time.time() - t0))
