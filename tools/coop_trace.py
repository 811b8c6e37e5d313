#!/usr/bin/env python
"""Phase times of svt_coop_kernel's workgroups (a -DSVT_COOP_TRACE=1 build prints them from the device):
    SVTYPER_HIP_LIB=svtyper_amd/csrc/variants/lib_trace.so python tools/coop_trace.py <units>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from svtyper_amd import hip, synth, evidence as ev
lib = bench.fixture_library()
n = int(sys.argv[1])
b = synth.make_units(n, 200, [lib], svtype_mix=(0.7, 0.15, 0.15, 0.0))
with hip.DeviceBatch(b, 0, ev.FLAG_RESULT96) as d:
    d.genotype(sync=True)
    print("---- second pass", flush=True)
    d.genotype(sync=True)
