#!/usr/bin/env bash
# headline (1 M units) and large_batch (4 M) kernel times for the default build and every variant
for lib in "" svtyper_amd/csrc/variants/lib_*.so; do
  [ -e "${lib:-/}" ] || continue
  if [ -n "$lib" ]; then export SVTYPER_HIP_LIB=$PWD/$lib; else unset SVTYPER_HIP_LIB; fi
  python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('${lib:-default}', '1M kernel_ms=%.4f frac=%.3f'%(d['roofline']['kernel_ms'], d['roofline']['frac']), '4M kernel_ms=%.4f frac=%.3f'%(d['large_batch']['kernel_ms'], d['large_batch']['frac']))"
done
