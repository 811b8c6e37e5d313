import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from svtyper_amd import hip
import ctypes
rt = ctypes.CDLL('libamdhip64.so')
n = 1_000_000
batch = bench.generate("c3_mixed_1m", n, 0, bench.usable_cpus())
for it in range(4):
    t0 = time.perf_counter(); d = hip.DeviceBatch(batch, 0, 0); t1 = time.perf_counter()
    if it >= 2: time.sleep(0.2)
    ts = time.perf_counter(); rt.hipDeviceSynchronize(); print('   hipDeviceSynchronize after create: %.2f ms' % ((time.perf_counter() - ts) * 1e3))
    ta = time.perf_counter(); ev_ms = d.genotype_timed(1); tb = time.perf_counter()
    d.genotype(sync=True); tc = time.perf_counter()
    r = d.results(); t3 = time.perf_counter(); d.close()
    print("iter %d: create %.1f ms | first pass: events %.3f ms, wall %.2f ms | second pass wall %.2f ms | results %.1f ms" % (
        it, (t1 - t0) * 1e3, ev_ms, (tb - ta) * 1e3, (tc - tb) * 1e3, (t3 - tc) * 1e3))
