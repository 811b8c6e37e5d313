python -m pytest tests/test_packed_evidence.py tests/test_result96.py tests/test_bench_contract.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -8
python bench.py --steps 10 --warmup 2 --legs packed,one_shot,c5,c5x --no-cpu-baseline > gpurun_out/r04_bench_c.json 2> gpurun_out/r04_bench_c.err; tail -3 gpurun_out/r04_bench_c.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_c.json'))
p=d['one_shot_packed']; print({k:round(p[k],2) for k in p if k.endswith('ms') or 'median' in k or 'equal' in k})
print('one_shot', d['one_shot']['wall_ms'])
print(d['c5_multisample'].get('one_shot'))
PY
SVT_TRACE=1 python - <<'PY' 2>&1 | grep -v "^\[svt\] pack: worker" | tail -40
import bench, time
from svtyper_amd import hip
b = bench.generate("c3_mixed_1m", 1000000, 0, 16)
out = hip.pinned_results(b.n_units)
for i in range(3):
    t=time.perf_counter(); hip.genotype_packed_from_records(b, 0, 0, out=out); print("call", i, (time.perf_counter()-t)*1e3, flush=True)
PY
