set -x
python -m pytest tests/test_hip_parity.py tests/test_hip_golden.py tests/test_packed_evidence.py tests/test_seams.py -m gpu -x -q 2>&1 | tail -5
SVT_TRACE=1 python tools/ab_stream.py 1000000 0 2>&1 | grep -v "^\[svt\] [a-z ]*  *[0-9.]* ms" | tail -20
python tools/ab_stream.py 1000000 1 2>&1 | tail -8
SVT_STREAM_WGS_PER_CU=3 AB_ROUNDS=1 python tools/ab_stream.py 1000000 0 NONE 2>&1 | tail -2
WORKLOADS=c5_multisample bash tools/ab_probe.sh 2>&1 | tail -6
