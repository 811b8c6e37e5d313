#!/usr/bin/env bash
one() { python bench.py --no-cpu-baseline --no-extra-legs --steps 10 "$@" 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$*', 'kernel_ms=%.4f frac=%.3f'%(d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
one; one --sso; one --workload c5_multisample; one --workload c5_multisample --sso
