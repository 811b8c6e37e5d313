#!/usr/bin/env bash
# Stall diagnosis of the genotype kernel: PMC passes only (no tracing).  Usage: tools/profile_stalls.sh <tag>
set -u
TAG=${1:-stalls}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense-leg"
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d $OUT/p1 -o pmc -- $B > /dev/null 2> $OUT/p1.err
# (a pass with the TCP_* latency / UTCL1 counters hung rocprofv3 on this pool until the outer timeout -- left out)
timeout 240 rocprofv3 --pmc MeanOccupancyPerCU OccupancyPercent MemUnitStalled LDSBankConflict -d $OUT/p3 -o pmc -- $B > /dev/null 2> $OUT/p3.err
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH -d $OUT/p4 -o pmc -- $B > /dev/null 2> $OUT/p4.err
timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE -d $OUT/p5 -o pmc -- $B > /dev/null 2> $OUT/p5.err
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
for sub in ("p1", "p3", "p4", "p5"):
    for f in sorted(glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True)):
        c = sqlite3.connect(f)
        q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
             "where kernel_name like '%svt_%' group by kernel_name, counter_name")
        for r in c.execute(q):
            print(sub, r[0][:60], r[1], "%.6g" % r[2], r[3])
PY
tail -3 $OUT/p3.err
