"""pass time vs number of units over the SAME record / result buffers (those of a 1.6 M-unit batch): round quantisation without the placement lottery"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from svtyper_amd import hip, evidence as ev
lib = hip.load()
lib.svt_debug_records_ptr.restype = C.c_void_p
lib.svt_debug_records_ptr.argtypes = [C.c_void_p]
lib.svt_debug_bind_records.argtypes = [C.c_void_p, C.c_void_p]
big = bench.generate("c3_mixed_1m", 1_600_000, 0, bench.usable_cpus())
D = hip.DeviceBatch(big, 0, ev.FLAG_RESULT96)
D.genotype(sync=True)
rec_ptr, res_ptr = lib.svt_debug_records_ptr(D._h), D.device_results_ptr()
sizes = [262_144, 393_216, 500_000, 524_288, 560_000, 600_000, 700_000, 786_432, 900_000, 960_000, 1_000_000, 1_024_000, 1_048_576, 1_070_000, 1_100_000, 1_200_000, 1_310_720, 1_400_000, 1_500_000, 1_572_864, 1_600_000]
for rep in range(2):
    for n in sizes:
        b = big.slice(0, n)
        with hip.DeviceBatch(b, 0, ev.FLAG_RESULT96) as d:
            hip._check(lib.svt_debug_bind_records(d._h, C.c_void_p(rec_ptr)))
            d.bind_device_results(res_ptr)
            d.genotype(sync=True)
            bench.spin_up(d, 20)
            ms = sorted(d.genotype_timed(20) / 20 for _ in range(5))
            alg, _ = d.bytes()
            d.bind_device_results(0)
            hip._check(lib.svt_debug_bind_records(d._h, None))
        wgs = (n + 511) // 512
        print("n %8d  WGs %5d = %.3f rounds   %.4f ms (median %.4f)   %.4f ns/unit   frac %.3f" % (n, wgs, wgs / 1024, ms[0], ms[2], ms[0] * 1e6 / n, alg / (ms[0] * 1e-3) / 8e12), flush=True)
    print()
