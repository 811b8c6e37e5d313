#!/usr/bin/env python
"""svt_pack_evidence (host encoder of packed evidence) on the configs[2] workload with 1 / 8 / 16 worker threads
(SVT_PACK_THREADS), best of three with pauses (the box's cgroup schedules 16 CPUs per 100 ms: bursts right behind one
another get throttled).  SVT_PACK_SCALAR=1: the record-by-record form; SVT_PACK_SPREAD=1 / 2: workers pinned to an L3 group / to one CPU of it (default: the scheduler places them);
SVT_PACK_PROBE=1: read the records only; SVT_TRACE=1: stage and per-worker times.   python tools/pack_scale.py"""
import time, sys, os
sys.path.insert(0, os.getcwd())
import bench
from svtyper_amd import hip
b = bench.generate("c3_mixed_1m", 1000000, 0, bench.usable_cpus())
for nt in [int(x) for x in sys.argv[1:]] or (1, 16, 0):     # 0 = the library's own choice
    if nt:
        os.environ["SVT_PACK_THREADS"] = str(nt)
    else:
        os.environ.pop("SVT_PACK_THREADS", None)
    times = []
    for i in range(int(os.environ.get("PACK_REPS", "5"))):
        time.sleep(0.4)      # the box schedules 16 CPUs per 100 ms (cgroup quota): back-to-back bursts get throttled
        t0 = time.perf_counter(); p = hip.PackedEvidence.try_pack(b); dt = time.perf_counter() - t0; p.free()
        times.append(dt * 1e3)
    best = min(times)
    print("threads %3s  pack best %.1f ms, all: %s  (%.2f ns per record at the best)" % (
        nt or "lib", best, " ".join("%.1f" % t for t in times), best / b.n_records * 1e6), flush=True)
