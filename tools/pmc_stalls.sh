#!/usr/bin/env bash
# stall / FIFO counters of the streaming kernel on the workload tools/ab_stream.py saved under /tmp/ab_stream
export TMPDIR=/tmp
[ -n "${1:-}" ] && export SVTYPER_HIP_LIB=$PWD/$1
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VSKIPPED SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
    d=/tmp/st_$$; rm -rf $d
    AB_REPS=3 timeout -k 5 120 rocprofv3 --pmc $set -d $d -o pmc -- python tools/ab_stream.py --child ${FLAGS:-0} > /dev/null 2>&1
    python - "$d" <<'PY'
import glob, os, sqlite3, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    for r in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%svt_stream%' group by counter_name"):
        print("%-30s %16.1f  n=%d" % (r[0], r[1], r[2]))
PY
done
