#!/usr/bin/env bash
# FETCH_SIZE / TCC hit counters of the streaming kernel for the default build and every variant, on the workload
# tools/ab_stream.py saved under /tmp/ab_stream (run that first).   tools/fetch_size.sh
export TMPDIR=/tmp
for lib in "" svtyper_amd/csrc/variants/lib_*.so; do
  [ -e "${lib:-/}" ] || continue
  if [ -n "$lib" ]; then export SVTYPER_HIP_LIB=$PWD/$lib; else unset SVTYPER_HIP_LIB; fi
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d=/tmp/fs_$$; rm -rf $d
    AB_REPS=3 timeout -k 5 120 rocprofv3 --pmc $set -d $d -o pmc -- python tools/ab_stream.py --child 0 > /dev/null 2>&1
    python - "$d" "${lib:-default}" <<'PY'
import glob, os, sqlite3, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    for r in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%svt_stream%' group by counter_name"):
        print("%-28s %-14s %14.1f  n=%d" % (os.path.basename(sys.argv[2]), r[0], r[1], r[2]))
PY
  done
done
