#!/usr/bin/env python
"""What state does bench.py's timed region find the device in?  One resident headline batch (tuned like bench.py's), then
(a) idle 2 s -> spin-up of S ms -> 3 warm-up passes -> 20 timed passes, for S in 0 ... 320, twice; (b) after 2 s of idling,
groups of 10 passes back to back for ~0.5 s (the pass time as the device comes out of idle and stays loaded).
   python tools/timed_state_probe.py [result_candidates record_candidates]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from svtyper_amd import evidence as ev, hip
b = bench.generate("c3_mixed_1m", 1_000_000, 0, bench.usable_cpus())
_num = [a for a in sys.argv[1:] if not a.startswith("--")]
rc = (int(_num[0]), int(_num[1])) if len(_num) > 1 else (32, 8)
WITH_TORCH = "--torch" in sys.argv      # as bench.py runs: torch's HIP context alive, the result buffer viewed as a tensor
SHORT = "--short" in sys.argv           # the series after the audition only
if WITH_TORCH:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
with hip.DeviceBatch(b, 0, ev.FLAG_RESULT96) as d:
    d.genotype(sync=True)
    cold = d.genotype_timed(20) / 20
    bench.spin_up(d, 40.0)
    if rc[0] or rc[1]:
        r = d.tune_placement(*rc)
        print("cold %.4f ms; audition %s" % (cold, r), flush=True)
    # the bench's own sequence after the audition (idle 2.5 s, spin-up 40 ms, 3 warm-ups, 20 timed), then the same timed group
    # again and again with 0.3 s of idling between: does the pass flip between levels over seconds?
    time.sleep(2.5)
    if WITH_TORCH:
        view = d.device_results_tensor()
        torch.cuda.synchronize()
    series = []
    for k in range(40):
        bench.spin_up(d, 40.0)
        for _ in range(3):
            d.genotype(sync=False)
        d.genotype(sync=True)
        series.append("%.4f" % (d.genotype_timed(20) / 20))
        time.sleep(0.3)
    print("after the audition, every 0.35 s: spin-up 40 ms + 3 warm-ups + timed 20%s ->\n" % (" (torch context alive)" if WITH_TORCH else "") + " ".join(series), flush=True)
    if WITH_TORCH:
        del view
    if SHORT:
        sys.exit(0)
    for rep in range(1):
        for spin in (0, 5, 10, 20, 40, 80, 160, 320):
            time.sleep(2.0)
            n = bench.spin_up(d, float(spin)) if spin else 0
            for _ in range(3):
                d.genotype(sync=False)
            d.genotype(sync=True)
            a = d.genotype_timed(20) / 20
            c = d.genotype_timed(20) / 20
            print("idle 2 s, spin-up %3d ms (%4d passes): timed 20 -> %.4f ms, the next 20 -> %.4f" % (spin, n, a, c), flush=True)
    time.sleep(2.0)
    t0 = time.perf_counter()
    line = []
    while time.perf_counter() - t0 < 0.6:
        line.append("%.0f:%.4f" % ((time.perf_counter() - t0) * 1e3, d.genotype_timed(10) / 10))
    print("after 2 s idle, groups of 10 (ms since start : pass ms):\n" + " ".join(line), flush=True)
    for idle in (0.0, 0.1, 0.5, 1.0):
        bench.spin_up(d, 400.0)
        time.sleep(idle)
        bench.spin_up(d, 40.0)
        print("400 ms of passes, idle %.1f s, spin-up 40 ms: timed 20 -> %.4f ms" % (idle, d.genotype_timed(20) / 20), flush=True)
