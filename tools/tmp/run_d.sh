python - <<'PY' 2>&1 | tail -40
import bench, time, os
from svtyper_amd import hip
b = bench.generate("c3_mixed_1m", 1000000, 0, 16)
out = hip.pinned_results(b.n_units)
hip.genotype_packed_from_records(b, 0, 0, out=out)
for ru in ("1000000", "500000", "250000", "125000", "83334", "serial"):
    if ru == "serial": os.environ["SVT_PACKED_SERIAL"] = "1"
    else: os.environ["SVT_PACK_RANGE_UNITS"] = ru
    ts = []
    for i in range(8):
        time.sleep(0.4)
        t=time.perf_counter(); hip.genotype_packed_from_records(b, 0, 0, out=out); ts.append((time.perf_counter()-t)*1e3)
    ts.sort()
    print("range units %8s: best %.2f  median %.2f  worst %.2f ms" % (ru, ts[0], ts[len(ts)//2], ts[-1]), flush=True)
PY
