#!/usr/bin/env python
"""Stage times of the pipelined one-shot entry points (svt_genotype / svt_genotype_packed) on the headline workload;
run with SVT_TRACE=1 for the library's own stage timer, SVT_PIPE_MB=<n> for another piece size."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from svtyper_amd import hip
batch = bench.generate("c3_mixed_1m", 1_000_000, 0, bench.usable_cpus())
out = hip.pinned_results(batch.n_units)
import ctypes as C
L = hip.load()
cb = batch.as_c()
for it in range(4):
    t0 = time.perf_counter(); hip.genotype_batch(batch, out=out); t1 = time.perf_counter()
    rc = L.svt_genotype(C.byref(cb), C.c_void_p(out.ptr()), 0, 0); t2 = time.perf_counter()
    print("pipelined wall %.2f ms (python wrapper) %.2f ms (C call only)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
