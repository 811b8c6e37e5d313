python -m pytest tests/test_result96.py tests/test_hip_parity.py tests/test_hip_golden.py tests/test_packed_evidence.py tests/test_multi_device.py -m gpu -x -q 2>&1 | tail -8
for wl in c3 c396 sso96 c596; do python tools/ab_inproc.py $wl 1000000 2>&1 | tail -8; done
