python -m pytest tests/test_hip_parity.py tests/test_hip_golden.py tests/test_packed_evidence.py -m gpu -x -q 2>&1 | tail -3
SVT_TRACE=1 python tools/ab_inproc.py c3 1000000 NONE 2>&1 | grep "kernel budget"
for wl in c3 sso c5 c5site; do python tools/ab_inproc.py $wl 1000000 2>&1 | tail -8; done
