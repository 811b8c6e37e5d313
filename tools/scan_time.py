#!/usr/bin/env python
"""svt_batch_create + pass of the configs[4] shape (1 M units, 66 libraries) three ways: with the svt_unit.libs hints,
without them (windows read off the records by svt_window_scan_kernel at create), and with SVT_FLAG_GENERAL_TABLES; the
last create runs with SVT_TRACE=1 (stage times).   python tools/scan_time.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from svtyper_amd import hip, evidence as ev
b = bench.generate("c5_multisample", 1_000_000, 0, bench.usable_cpus())
hinted = b.units.copy()
for rep in range(3):
    for name, fl, libs0 in (("hinted", 0, False), ("derived", 0, True), ("general", ev.FLAG_GENERAL_TABLES, True)):
        b.units["libs"] = 0 if libs0 else hinted["libs"]
        t0 = time.perf_counter()
        d = hip.DeviceBatch(b, 0, fl)
        t1 = time.perf_counter()
        d.genotype(sync=True)
        t2 = time.perf_counter()
        ms = min(d.genotype_timed(5) / 5 for _ in range(3))
        print("rep %d %-8s mode %d create %.2f ms first pass %.3f ms pass %.4f ms" % (rep, name, d.table_mode(), (t1 - t0) * 1e3, (t2 - t1) * 1e3, ms), flush=True)
        d.close()
os.environ["SVT_TRACE"] = "1"
b.units["libs"] = 0
d = hip.DeviceBatch(b, 0, 0); d.close()
