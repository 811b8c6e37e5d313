#!/usr/bin/env python
"""Pass time of a 1 M-unit batch with three libraries and NO svt_unit.libs hints: the whole batch becomes one library
window when its tables fit LDS (table mode 1), else the general mode (2).   python tools/multilib_nohints.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from svtyper_amd import hip, synth, evidence as ev
lib = bench.fixture_library()
libs = [lib, synth.normal_library(420.0, 95.0, seed=3), synth.normal_library(280.0, 40.0, seed=4)]
parts = [synth.make_units(125_000, 100 + i, libs, svtype_mix=(0.7, 0.15, 0.15, 0.0)) for i in range(8)]
b = ev.concat_batches(parts)
with hip.DeviceBatch(b, 0, 0) as d:
    d.genotype(sync=True)
    ms = min(d.genotype_timed(10) / 10 for _ in range(3))
    alg, _ = d.bytes()
    print("3 libraries, no hints: mode", d.table_mode(), "pass %.4f ms frac %.3f" % (ms, alg / ms / 1e9 / 8))
