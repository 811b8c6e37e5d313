#!/usr/bin/env python
"""Condense rocprofv3 output dirs (tools/profile.sh) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


for f in find("stats/**/*kernel_stats.csv"):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("  %-60s calls=%s avg_ns=%s total_ns=%s pct=%s" % (
                row.get("Name", "")[:60], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"),
                row.get("Percentage")))

for d in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in find(d + "/**/*counter_collection.csv"):
        agg = defaultdict(lambda: defaultdict(list))
        extra = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                agg[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
                extra[k] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"),
                            row.get("LDS_Block_Size"), row.get("Grid_Size"), row.get("Workgroup_Size"))
        print("== counters (%s)" % os.path.relpath(f, out))
        for k, cs in agg.items():
            if "genotype" not in k and "repack" not in k:
                continue
            print("  kernel %s  vgpr/agpr/sgpr/lds/grid/wg=%s" % (k[:70], extra[k]))
            for c, v in sorted(cs.items()):
                print("    %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
