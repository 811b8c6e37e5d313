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


# ---- the entry bench.py reads back (profiles/hbm_traffic.json), stamped with the build it was measured on
import json
try:
    line = [l for l in open(os.path.join(out, "stats_bench.json")) if l.startswith("{")][-1]
    bj = json.loads(line)
    kern = bj["roofline"]["kernel"]
    vals = {}
    for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for f in dbs(sub):
            c = sqlite3.connect(f)
            r = c.execute("select avg(value) from counters_collection where kernel_name like ? and counter_name = ?",
                          ("%" + kern + "%", name)).fetchone()
            vals[name] = r[0]
    avg_us = None
    for f in dbs("stats"):
        c = sqlite3.connect(f)
        r = c.execute("select average from top_kernels where name like ?", ("%" + kern + "%",)).fetchone()
        if r:
            avg_us = r[0] / 1e3 if r[0] > 1e4 else r[0]
    layout = bj["config"]["device_layout"].split(",")[0]
    key = {"the canonical CSR records as uploaded": "stream"}.get(layout, layout)
    entry = {key: {
        "kernel": kern, "workload": bj["config"]["workload"], "units": bj["config"]["units_per_gpu"],
        "records": bj["config"]["records_per_gpu"], "FETCH_SIZE_KB": vals.get("FETCH_SIZE"), "WRITE_SIZE_KB": vals.get("WRITE_SIZE"),
        # gfx950: FETCH_SIZE reports half the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section)
        "traffic_bytes_per_launch": int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024),
        "kernel_trace_avg_us": avg_us, "library_sha16": bj["roofline"]["library_sha16"],
        "source": "tools/profile.sh %s" % os.path.basename(out.rstrip("/")),
    }}
    with open(os.path.join(out, "hbm_traffic_entry.json"), "w") as f:
        json.dump(entry, f, indent=1)
    print("# hbm_traffic entry:", json.dumps(entry))
except Exception as e:   # the text summary above is still good
    print("# no hbm_traffic entry:", repr(e))
