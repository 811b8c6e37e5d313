#!/usr/bin/env python
"""configs[4] shape: the same (site, sample) units in site-major order (unit = site * 32 + sample, what a joint
caller emits) and in sample-major order (all sites of sample 0, then sample 1, ...: what a producer that reads BAM
by BAM emits) through the library-window kernel.   python tools/c5_sample_major.py [units]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


if __name__ == "__main__":
    import bench
    from svtyper_amd import hip
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    b = bench.generate("c5_multisample", n, 0, bench.usable_cpus())
    from svtyper_amd import synth
    sm, order = synth.to_sample_major(b, bench.N_SAMPLES_C5)
    res = {}
    for name, batch in (("site-major", b), ("sample-major", sm)):
        with hip.DeviceBatch(batch, 0, 0) as d:
            if name == "sample-major":
                d.result_order(bench.N_SAMPLES_C5)
            d.genotype(sync=True)
            ms = sorted(d.genotype_timed(10) / 10 for _ in range(9))
            alg, _ = d.bytes()
            res[name] = d.results().rec
            print("%-13s table_mode %d  pass %.4f ms (median %.4f)  frac %.3f" % (name, d.table_mode(), ms[0], ms[4], alg / ms[0] / 1e6 / 8000), flush=True)
    print("site-major results from both orders equal:", bool(np.array_equal(res["site-major"], res["sample-major"])))
