"""equal workgroups in whole rounds (wg_plan) vs full workgroups, over the SAME record / result buffers"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from svtyper_amd import hip, evidence as ev, synth
lib = hip.load()
lib.svt_debug_records_ptr.restype = C.c_void_p
lib.svt_debug_records_ptr.argtypes = [C.c_void_p]
lib.svt_debug_bind_records.argtypes = [C.c_void_p, C.c_void_p]
lib.svt_debug_wg_balance.argtypes = [C.c_int]
lib.svt_debug_device_alloc.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
_p = C.c_void_p()
hip._check(lib.svt_debug_device_alloc(0, 3_300_000 * 128, 0, C.byref(_p)))
RES = int(_p.value)        # one result buffer for every batch of this run

def run(label, batch, flags, sizes, order=0):
    lib.svt_debug_wg_balance(0)
    D = hip.DeviceBatch(batch, 0, flags)
    if order: D.result_order(order)
    D.genotype(sync=True)
    want = D.results().rec.tobytes() if batch.n_units <= 2_100_000 else None
    rec_ptr = lib.svt_debug_records_ptr(D._h)
    for n in sizes:
        b = batch if n == batch.n_units else batch.slice(0, n)
        row = []
        for bal in FILLS:
            lib.svt_debug_wg_balance(bal)
            with hip.DeviceBatch(b, 0, flags) as d:
                if order: d.result_order(order)
                hip._check(lib.svt_debug_bind_records(d._h, C.c_void_p(rec_ptr)))
                slots = d.result_slots()
                d.bind_device_results(RES)
                d.genotype(sync=True)
                same = (d.results().rec.tobytes() == want) if (n == batch.n_units and want is not None) else None
                bench.spin_up(d, 20)
                ms = sorted(d.genotype_timed(20) / 20 for _ in range(5))
                alg, _ = d.bytes()
                hip._check(lib.svt_debug_bind_records(d._h, None))
                d.bind_device_results(0)
            row.append("%s %.4f (%.3f) slots %d%s" % (("fill>=%d" % bal) if bal else "full    ", ms[0], alg / (ms[0] * 1e-3) / 8e12, slots, "" if same is None else (" equal" if same else " DIFFER")))
        print("%s n %8d: %s" % (label, n, " | ".join(row)), flush=True)
    D.close()

FILLS = (0, 75, 66, 50, 34)
big = bench.generate("c3_mixed_1m", 2_300_000, 0, bench.usable_cpus())
run("classic r96", big, ev.FLAG_RESULT96, [560_000, 600_000, 700_000, 1_070_000, 1_100_000, 1_400_000, 1_700_000, 2_200_000, 2_300_000])
run("classic 128", big, 0, [600_000, 1_100_000])
