#!/usr/bin/env bash
# Run on the GPU box (through gpurun): kernel-trace stats + PMC passes for the dominant kernel of bench.py.
# Usage: tools/profile.sh <tag>    -> gpurun_out/prof_<tag>/{summary.txt, hbm_traffic_entry.json, stats_bench.json}
# Copy summary.txt to profiles/<round>_<tag>_rocprof.txt and merge hbm_traffic_entry.json into
# profiles/hbm_traffic.json with tools/update_traffic.py: bench.py only accepts an entry measured on the very
# kernel sources it is running on (source_sha16: csrc/* + include/*, so a rebuild keeps the entry).
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the headline plus the sso and configs[4]-shape legs: each has its own kernel instantiation, so the per-kernel PMC means stay apart
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --legs ${LEGS:-sso,c5} ${BENCH_ARGS:-}"
BENCH_SHORT="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs ${LEGS:-sso,c5} ${BENCH_ARGS:-}"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/stats_bench.json 2> $OUT/stats.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmc_sq1 -o pmc -- $BENCH_SHORT > /dev/null 2> $OUT/pmc_sq1.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d $OUT/pmc_sq2 -o pmc -- $BENCH_SHORT > /dev/null 2> $OUT/pmc_sq2.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH_SHORT > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH_SHORT > /dev/null 2> $OUT/pmc_write.err
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
