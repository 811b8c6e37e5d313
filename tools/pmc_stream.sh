#!/usr/bin/env bash
# PMC passes over the streaming kernel on the workload tools/ab_stream.py saved under /tmp/ab_stream (run that first).
# Usage: tools/pmc_stream.sh <tag> [lib.so]   -> gpurun_out/pmc_<tag>.txt
set -u
TAG=${1:-x}; LIB=${2:-}
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$LIB" ] && export SVTYPER_HIP_LIB=$LIB
C="python tools/ab_stream.py --child ${FLAGS:-0}"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "MeanOccupancyPerCU"; do
    d=$OUT/$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --pmc $set -d $d -o pmc -- $C > /dev/null 2> $d.err
done
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $C > /dev/null 2> $OUT/stats.err
python - "$OUT" <<'PY' | tee $OUT.txt
import glob, os, sqlite3, sys
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "**", "*.db"), recursive=True)):
    c = sqlite3.connect(f)
    try:
        for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%svt_%' group by kernel_name, counter_name"):
            print("%-60s %-24s %14.6g  n=%d" % (r[0][:60], r[1], r[2], r[3]))
    except sqlite3.OperationalError:
        pass
    try:
        for r in c.execute("select name,total_calls,total_duration,average from top_kernels"):
            print("stats: %-60s calls %d total_us %.1f avg_us %.2f" % (r[0][:60], r[1], r[2] / 1e3 if r[2] > 1e6 else r[2], r[3] / 1e3 if r[3] > 1e4 else r[3]))
    except sqlite3.OperationalError:
        pass
PY
