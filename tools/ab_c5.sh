#!/usr/bin/env bash
# kernel time of the configs[4] shape for every library variant under svtyper_amd/csrc/variants (and the default build)
for lib in "" svtyper_amd/csrc/variants/lib_*.so; do
  if [ -n "$lib" ]; then export SVTYPER_HIP_LIB=$PWD/$lib; else unset SVTYPER_HIP_LIB; fi
  python bench.py --workload c5_multisample --no-cpu-baseline --no-extra-legs --steps 10 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('${lib:-default}', 'kernel_ms=%.4f'%d['roofline']['kernel_ms'], 'frac=%.3f'%d['roofline']['frac'])"
done
