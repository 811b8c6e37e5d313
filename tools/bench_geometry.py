#!/usr/bin/env python
"""Geometry stage at scale: the fake-site fragment summaries replicated to ~N fragments, device
(svt_geometry_kernel inside svt_batch_create_from_fragments) vs the host packer (packer.pack_fragments).
Run with SVT_TRACE=1 for the stage times; wrap in rocprofv3 --kernel-trace --stats for the kernel time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import goldenio as gio, fakereads
from svtyper_amd import geometry as geo, fragments as fr, packer, hip, evidence as ev

target = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
g = gio.load("fake_sites.json.gz")
grp = g["groups"][0]
class L: pass
libs = []
for x in grp["libraries"]:
    l = L(); l.name = x["name"]; l.mean = gio.fh(x["mean"]); l.sd = gio.fh(x["sd"]); libs.append(l)
rg = {r: l for l, x in zip(libs, grp["libraries"]) for r in x["readgroups"]}
idx = {id(l): i for i, l in enumerate(libs)}
tid_of = lambda c: {"1": 0, "2": 1}.get(c, -1)
sites = []
for s in grp["sites"]:
    fs = {}
    for t in s["reads"]:
        r = fakereads.FakeRead(*t)
        if r.query_name in fs: fs[r.query_name].add_read(r)
        else: fs[r.query_name] = fr.SamFragment(r, rg[r.get_tag("RG")])
    sites.append((s["breakpoint"], fs))
t0 = time.perf_counter(); nrec = 0
for bp, fs in sites: nrec += len(packer.pack_fragments(fs, bp, idx, 20, 3))
host_s = time.perf_counter() - t0
print("host packer (Python): %d fragments in %.3f s = %.0f fragments/s" % (nrec, host_s, nrec / host_s))
t0 = time.perf_counter()
b = geo.FragmentBatchBuilder(gio.libraries(grp["libraries"]), 1.0, 1.0, 20, 3)
for bp, fs in sites: b.add(geo.breakpoint_record(bp, tid_of), geo.summarise_fragments(fs, bp, idx, tid_of))
base = b.build()
print("host summariser (Python): %.0f fragments/s" % (base.n_fragments / (time.perf_counter() - t0)))
rep = max(1, target // base.n_fragments)
off = np.concatenate([[0]] + [base.frag_offset[1:].astype(np.int64) + k * base.n_fragments for k in range(rep)]).astype(np.uint64)
big = geo.FragmentBatch(off, np.tile(base.breakpoints, rep), np.tile(base.fragments, rep), base.libs, 1.0, 1.0, 20, 3)
print("device batch: %d units, %d fragments (%.1f MB of summaries)" % (big.n_units, big.n_fragments, big.n_fragments * 128 / 1e6))
for it in range(3):
    t0 = time.perf_counter(); d = hip.DeviceBatch.from_fragments(big, 0, 0); t1 = time.perf_counter(); d.close()
    print("create_from_fragments: %.1f ms = %.1f M fragments/s end to end" % ((t1 - t0) * 1e3, big.n_fragments / (t1 - t0) / 1e6))
