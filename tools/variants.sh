#!/usr/bin/env bash
# A/B kernel build variants on the GPU box: tools/variants.sh  (libs under svtyper_amd/csrc/variants)
for f in svtyper_amd/csrc/variants/lib_*.so; do
  for rep in 1 2; do
    SVTYPER_HIP_LIB=$PWD/$f python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$f', 'kernel_ms=%.4f'%d['roofline']['kernel_ms'], 'frac=%.3f'%d['roofline']['frac'], 'value=%.3e'%d['value'])"
  done
done
