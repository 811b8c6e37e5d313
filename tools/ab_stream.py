#!/usr/bin/env python
"""A/B timing of library build variants (svtyper_amd/csrc/variants/lib_*.so, see tools/stream_variants.sh) on one
workload generated once:   python tools/ab_stream.py [units] [flags]
Each variant runs in its own process (SVTYPER_HIP_LIB), prints the pass time and a digest of the result records."""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

TMP = "/tmp/ab_stream"


def child(flags):
    import bench
    from svtyper_amd import evidence as ev, hip
    batch = ev.EvidenceBatch(np.load(TMP + "/off.npy"), np.load(TMP + "/units.npy"), np.load(TMP + "/recs.npy"),
                             [bench.fixture_library()])
    with hip.DeviceBatch(batch, 0, flags) as d:
        d.genotype(sync=True)
        ms = sorted(d.genotype_timed(10) / 10 for _ in range(int(os.environ.get("AB_REPS", "15"))))
        alg, _ = d.bytes()
        dig = hashlib.sha1(d.results().rec.tobytes()).hexdigest()[:12]
    print("%-34s pass %.4f ms (median %.4f)  %.0f GB/s alg  frac %.3f  digest %s" % (
        os.path.basename(os.environ.get("SVTYPER_HIP_LIB", "default")), ms[0], ms[len(ms) // 2], alg / ms[0] / 1e6,
        alg / ms[0] / 1e6 / 8000, dig), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
        sys.exit(0)
    import bench
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = sys.argv[3:]   # variant name filters
    os.makedirs(TMP, exist_ok=True)
    t0 = time.time()
    batch = bench.generate("c3_mixed_1m", n, 0, bench.usable_cpus())
    np.save(TMP + "/off.npy", batch.rec_offset); np.save(TMP + "/units.npy", batch.units); np.save(TMP + "/recs.npy", batch.records)
    print("generated %d units in %.1f s" % (n, time.time() - t0), flush=True)
    import glob
    libs = sorted(glob.glob(os.path.join(ROOT, "svtyper_amd", "csrc", "variants", "lib_*.so")))
    if only:
        libs = [l for l in libs if any(o in os.path.basename(l) for o in only)]
    for lib in ([None] + libs) * int(os.environ.get("AB_ROUNDS", "2")):
        env = dict(os.environ)
        if lib:
            env["SVTYPER_HIP_LIB"] = lib
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child", str(flags)], env=env)
