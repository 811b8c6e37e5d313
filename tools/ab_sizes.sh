#!/usr/bin/env bash
# kernel time by batch size for the default build and every variant under svtyper_amd/csrc/variants
one() { python bench.py --no-cpu-baseline --no-extra-legs --steps 10 "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('${SVTYPER_HIP_LIB:-default}', '$*', 'kernel_ms=%.4f frac=%.3f'%(d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for n in ${SIZES:-100000 196608 250000 500000 1000000 2000000}; do
  for lib in "" svtyper_amd/csrc/variants/lib_*.so; do
    [ -e "${lib:-/}" ] || continue
    if [ -n "$lib" ]; then export SVTYPER_HIP_LIB=$PWD/$lib; else unset SVTYPER_HIP_LIB; fi
    one --units $n ${EXTRA_ARGS:-}
  done
done
