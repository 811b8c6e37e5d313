#!/usr/bin/env bash
# LDS bank-conflict share of the streaming kernel (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) for the default build and
# every variant, on the workload tools/ab_stream.py saved under /tmp/ab_stream (run that first).
export TMPDIR=/tmp
for lib in "" svtyper_amd/csrc/variants/lib_*.so; do
  [ -e "${lib:-/}" ] || continue
  if [ -n "$lib" ]; then export SVTYPER_HIP_LIB=$PWD/$lib; else unset SVTYPER_HIP_LIB; fi
  d=/tmp/lc_$$; rm -rf $d
  AB_REPS=3 timeout -k 5 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $d -o pmc -- python tools/ab_stream.py --child ${FLAGS:-0} > /dev/null 2>&1
  python - "$d" "${lib:-default}" <<'PY'
import glob, os, sqlite3, sys
v = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    for r in c.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%svt_stream%' group by counter_name"):
        v[r[0]] = r[1]
if v:
    print("%-24s conflicts %.1f %% of LDS cycles (%.1f M of %.1f M); LDS instrs %.1f M, VALU %.1f M, issue-stalled %.1f %% of wave cycles" % (
        os.path.basename(sys.argv[2]), 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], v["SQ_LDS_BANK_CONFLICT"] / 1e6, v["SQ_LDS_IDX_ACTIVE"] / 1e6,
        v["SQ_INSTS_LDS"] / 1e6, v["SQ_INSTS_VALU"] / 1e6, 100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
done
