import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
from svtyper_amd import synth
from oracle import c_oracle
sys.path.insert(0, ROOT)
import bench
b = synth.make_units(200000, 5, [bench.fixture_library()])
import numpy as np
for nt in (1, 8, 16, 32, 64, 128, 256):
    c_oracle.genotype_batch(b, 0, nt)
    t0 = time.perf_counter(); c_oracle.genotype_batch(b, 0, nt); dt = time.perf_counter() - t0
    print("oracle threads %3d: %.3f s = %.2f M units/s" % (nt, dt, 0.2 / dt))
