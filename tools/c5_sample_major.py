#!/usr/bin/env python
"""configs[4] shape: the same (site, sample) units in site-major order (unit = site * 32 + sample, what a joint
caller emits) and in sample-major order (all sites of sample 0, then sample 1, ...: what a producer that reads BAM
by BAM emits) through the library-window kernel.   python tools/c5_sample_major.py [units]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def sample_major(batch, n_samples):
    from svtyper_amd import evidence as ev
    n = batch.n_units
    n_sites = n // n_samples
    order = (np.arange(n_sites, dtype=np.int64)[None, :] * n_samples + np.arange(n_samples, dtype=np.int64)[:, None]).reshape(-1)
    off = batch.rec_offset.astype(np.int64)
    cnt = (off[1:] - off[:-1])[order]
    new_off = np.zeros(n + 1, np.uint64)
    new_off[1:] = np.cumsum(cnt)
    src = np.repeat(off[:-1][order] - new_off[:-1].astype(np.int64), cnt) + np.arange(int(new_off[-1]), dtype=np.int64)
    return ev.EvidenceBatch(new_off, batch.units[order], batch.records[src], batch.libs, batch.split_weight, batch.disc_weight), order


if __name__ == "__main__":
    import bench
    from svtyper_amd import hip
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    b = bench.generate("c5_multisample", n, 0, bench.usable_cpus())
    sm, order = sample_major(b, bench.N_SAMPLES_C5)
    res = {}
    for name, batch in (("site-major", b), ("sample-major", sm)):
        with hip.DeviceBatch(batch, 0, 0) as d:
            d.genotype(sync=True)
            ms = sorted(d.genotype_timed(10) / 10 for _ in range(9))
            alg, _ = d.bytes()
            res[name] = d.results().rec
            print("%-13s table_mode %d  pass %.4f ms (median %.4f)  frac %.3f" % (name, d.table_mode(), ms[0], ms[4], alg / ms[0] / 1e6 / 8000), flush=True)
    print("results equal after reordering:", bool(np.array_equal(res["site-major"][order], res["sample-major"])))
