#!/usr/bin/env bash
# kernel time of a workload for the default build and every variant under svtyper_amd/csrc/variants
run() { # lib workload
  if [ -n "$1" ]; then export SVTYPER_HIP_LIB=$PWD/$1; else unset SVTYPER_HIP_LIB; fi
  python bench.py --workload $2 --no-cpu-baseline --no-extra-legs --steps 10 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('${1:-default}', '$2', '${SVT_BENCH_C5_LIBS:-}', 'kernel_ms=%.4f'%d['roofline']['kernel_ms'], 'frac=%.3f'%d['roofline']['frac'])"
}
for wl in ${WORKLOADS:-c3_mixed_1m c5_multisample}; do
  for lib in "" svtyper_amd/csrc/variants/lib_*.so; do [ -e "${lib:-/}" ] && run "$lib" $wl; done
done
