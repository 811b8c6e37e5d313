#!/usr/bin/env python
"""Pass time of the general mode (tables through L2): the configs[4] shape with the svt_unit.libs hints removed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from svtyper_amd import hip, evidence as ev
b = bench.generate("c5_multisample", 500_000, 0, bench.usable_cpus())
b.units["libs"] = 0
with hip.DeviceBatch(b, 0, ev.FLAG_GENERAL_TABLES if os.environ.get("GENERAL", "1") == "1" else 0) as d:   # GENERAL=0: windows derived from the records
    d.genotype(sync=True)
    ms = min(d.genotype_timed(5) / 5 for _ in range(3))
    alg, _ = d.bytes()
    print(os.path.basename(os.environ.get("SVTYPER_HIP_LIB", "default")), "mode", d.table_mode(), "units", b.n_units, "pass %.4f ms frac %.3f" % (ms, alg / ms / 1e9 / 8))
