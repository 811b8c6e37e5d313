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
# one rocprofv3 pass; a pass that hangs (seen: a counter pass that never returns) is killed after 150 s and tried once more
run_pass() {   # run_pass <subdir> <stdout file> <bench command> -- <rocprofv3 options...>
    local sub=$1 outf=$2 cmd=$3; shift 4
    for attempt in 1 2; do
        rm -rf $OUT/$sub
        if timeout -k 10 150 rocprofv3 "$@" -d $OUT/$sub -o ${sub%%_*} -- $cmd > $outf 2> $OUT/$sub.err; then return 0; fi
        echo "pass $sub attempt $attempt failed" >> $OUT/passes.log
    done
    return 1
}
# A process of its own for the HEADLINE's timed region, in the state the bench line is quoted in (svt_batch_tune_placement first,
# as the default run; HEAD_TUNE="--tune-placement 0,0" for the untuned state): no other leg, so the timed region is the LAST `steps`
# dispatches of the headline kernel (roofline.timed_region_dispatches.from_end = 0) and the trace and the counter passes are
# averaged over exactly the launches bench.py times (tools/summarize_prof.py, "timed region")
HEAD="python bench.py --steps ${HEAD_STEPS:-50} --warmup 3 --no-cpu-baseline --no-extra-legs ${HEAD_TUNE:-} ${BENCH_ARGS:-}"
run_pass headstats $OUT/stats_headline.json "$HEAD" -- --kernel-trace --stats
run_pass headfetch $OUT/fetch_headline.json "$HEAD" -- --pmc FETCH_SIZE
run_pass headwrite $OUT/write_headline.json "$HEAD" -- --pmc WRITE_SIZE
run_pass stats $OUT/stats_bench.json "$BENCH" -- --kernel-trace --stats
run_pass pmc_sq1 /dev/null "$BENCH_SHORT" -- --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run_pass pmc_sq2 /dev/null "$BENCH_SHORT" -- --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run_pass pmc_fetch /dev/null "$BENCH_SHORT" -- --pmc FETCH_SIZE
run_pass pmc_write /dev/null "$BENCH_SHORT" -- --pmc WRITE_SIZE
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the raw rocpd databases are tens of MB (gpurun copies at most 64 MiB back): keep the summary, the bench lines and the logs
[ -n "${KEEP_DBS:-}" ] || find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
