#!/usr/bin/env bash
# R = 2 candidate: the gpu tests on the variant library, then headline / sso / configs[4] kernel times for both builds
export SVTYPER_HIP_LIB=$PWD/svtyper_amd/csrc/variants/lib_r2.so
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
unset SVTYPER_HIP_LIB
one() { python bench.py --no-cpu-baseline --no-extra-legs --steps 10 "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('${SVTYPER_HIP_LIB:-default}', '$*', 'kernel_ms=%.4f frac=%.3f'%(d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for lib in "" svtyper_amd/csrc/variants/lib_r2.so; do
  if [ -n "$lib" ]; then export SVTYPER_HIP_LIB=$PWD/$lib; else unset SVTYPER_HIP_LIB; fi
  one --sso; one --workload c5_multisample; one --workload c2_del_100k; one --units 250000; one --units 2000000
done
