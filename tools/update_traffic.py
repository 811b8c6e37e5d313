#!/usr/bin/env python
"""Merge gpurun_out/prof_<tag>/hbm_traffic_entry.json (tools/profile.sh) into profiles/hbm_traffic.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
entry = json.load(open(sys.argv[1]))
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
cur = json.load(open(path)) if os.path.exists(path) else {}
cur.update(entry)
json.dump(cur, open(path, "w"), indent=1)
print("updated", path, "with", list(entry))
