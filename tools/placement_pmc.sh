#!/usr/bin/env bash
# tools/placement_probe under rocprofv3 --pmc: translation-cache counters per buffer (17 dispatches each), beside the probe's own times
export TMPDIR=/tmp
d=/tmp/pp_$$; rm -rf $d
timeout -k 5 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY -d $d -o pmc -- ./tools/placement_probe ${1:-6} 2>&1 | sed -E "s/: wr .* spin +150: / /" | grep -E "buffer|slice"
python3 - "$d" <<'PY'
import glob, os, sqlite3, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    key = "dispatch_id" if "dispatch_id" in cols else cols[0]
    rows = list(c.execute("select %s, counter_name, value from counters_collection where kernel_name like '%%gather%%' order by %s" % (key, key)))
    by = {}
    for d, n, v in rows:
        by.setdefault(d, {})[n] = by.get(d, {}).get(n, 0) + v
    ids = sorted(by)
    print("dispatches", len(ids), "columns", cols)
    for g in range(0, len(ids), 17):
        grp = ids[g:g + 17]
        names = sorted(by[grp[0]])
        print("dispatch group %2d:" % (g // 17), "  ".join("%s %.3e" % (n, sum(by[i].get(n, 0) for i in grp) / len(grp)) for n in names))
PY
