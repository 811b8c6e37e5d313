#!/usr/bin/env python
"""Corrupted BAMs (bit flips, truncation, overwritten spans) through the native reader, one subprocess each: the
reader must answer with an error (or a result), never crash."""
import os, sys, subprocess, tempfile, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
child = r'''
import sys, json, os
ROOT = os.environ["SVT_ROOT"]; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_native_reads as N
from svtyper_amd import bam, library, native_reads as nr, hip
path = sys.argv[1]
info = json.load(open(sys.argv[2]))
sites = json.load(open(sys.argv[3]))
try:
    nb = nr.NativeBam(path)
    pybam = bam.AlignmentFile(sys.argv[4])           # the intact copy, for the library / windows only
    sample = library.Sample.from_lib_info(pybam, info, 1e-3)
    N._native_summaries(sites, sample, nb, nr.COUNT_CLASSIC, None, 2)
    print("ok")
except hip.SvtyperHipError as e:
    print("error:", str(e)[:80])
'''
tmp = tempfile.mkdtemp()
import test_native_reads as N, json
good = os.path.join(tmp, "good.bam")
sites, info = N._synthetic_bam(good, seed=77, n_pairs=300)
json.dump(info, open(os.path.join(tmp, "info.json"), "w")); json.dump(sites, open(os.path.join(tmp, "sites.json"), "w"))
raw = open(good, "rb").read(); bai = open(good + ".bai", "rb").read()
rng = random.Random(5)
outcomes = {}
for it in range(int(os.environ.get('SVT_FUZZ_ITERS', '60'))):
    b = bytearray(raw)
    mode = it % 3
    if mode == 0:
        for _ in range(rng.randint(1, 8)): b[rng.randrange(200, len(b))] ^= 1 << rng.randrange(8)
    elif mode == 1:
        b = b[: rng.randrange(300, len(b))]
    else:
        p = rng.randrange(200, len(b) - 64); b[p:p + 32] = bytes(rng.randrange(256) for _ in range(32))
    bad = os.path.join(tmp, "bad%d.bam" % it)
    open(bad, "wb").write(bytes(b)); open(bad + ".bai", "wb").write(bai)
    r = subprocess.run([sys.executable, "-c", child, bad, os.path.join(tmp, "info.json"), os.path.join(tmp, "sites.json"), good],
                       capture_output=True, text=True, timeout=120, env=dict(os.environ, SVT_ROOT=ROOT))
    key = "crash rc=%d" % r.returncode if r.returncode != 0 else r.stdout.strip().split(":")[0]
    outcomes[key] = outcomes.get(key, 0) + 1
    if r.returncode != 0: print(it, mode, r.stderr[-300:])
print(outcomes)
if any(k.startswith("crash") for k in outcomes):
    sys.exit(1)
