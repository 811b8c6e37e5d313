#!/usr/bin/env python
"""Condense the rocprofv3 output of tools/profile.sh (rocpd sqlite databases) into a text summary
that is small enough to commit under profiles/."""
import glob
import os
import sqlite3
import sys

out = sys.argv[1]


def dbs(sub):
    return sorted(glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True))


for f in dbs("stats"):
    c = sqlite3.connect(f)
    print("# rocprofv3 --kernel-trace --stats   (%s)" % os.path.relpath(f, out))
    print("name,total_calls,total_duration_us,average_us,percentage")
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(",".join(str(x) for x in r))

for sub in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in dbs(sub):
        c = sqlite3.connect(f)
        print("# rocprofv3 --pmc pass %s   (per-dispatch mean over the svt_* kernels)" % sub)
        print("kernel,counter,mean,dispatches,vgpr_count,lds_block_size,sgpr_count,grid,workgroup")
        q = ("select kernel_name, counter_name, avg(value), count(*), max(vgpr_count), max(lds_block_size), "
             "max(sgpr_count), max(grid_size), max(workgroup_size) from counters_collection "
             "where kernel_name like '%svt_%' group by kernel_name, counter_name")
        for r in c.execute(q):
            name = r[0].replace("(anonymous namespace)::", "")
            print(",".join([name] + [str(x) for x in r[1:]]))
