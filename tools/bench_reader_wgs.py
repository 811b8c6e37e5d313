#!/usr/bin/env python
"""Native reader on a BAM that looks like whole-genome sequencing: 150-bp pairs at ~30x over a few Mbp with random
bases and binned qualities (so BGZF blocks inflate at a realistic cost), 64-KiB blocks, one DEL site every few kb --
every site touches blocks nobody has inflated yet, unlike the repeated-site benchmarks.  CPU only."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bamwriter as bw
import test_native_reads as N
from svtyper_amd import bam, library, native_reads as nr

GENOME = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
COV = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
THREADS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 4, 15]
SPACING = int(sys.argv[4]) if len(sys.argv) > 4 else 4_000
rng = np.random.default_rng(1)
n_pairs = int(GENOME * COV / 300)
tmp = os.environ.get("WGS_DIR") or tempfile.mkdtemp()      # WGS_DIR: keep the BAM between runs
os.makedirs(tmp, exist_ok=True)
path = os.path.join(tmp, "wgs_%d_%g.bam" % (GENOME, COV))
header = "@HD\tVN:1.5\tSO:coordinate\n@SQ\tSN:1\tLN:%d\n@RG\tID:rg\tSM:smp\tLB:lib\n" % GENOME
t0 = time.time()
starts = rng.integers(0, GENOME - 1000, n_pairs)
isz = np.clip(rng.normal(400, 60, n_pairs), 160, 900).astype(np.int64)
seq_pool = rng.integers(0, 4, (4096, 75)); seq_pool = ((1 << seq_pool) << 4 | (1 << rng.integers(0, 4, (4096, 75)))).astype(np.uint8)
qual_vals = np.array([2, 11, 25, 37], np.uint8)
qual_pool = qual_vals[rng.choice(4, (4096, 150), p=[0.02, 0.08, 0.2, 0.7])]
recs = []
mqs = [0, 20, 37, 60, 60, 60, 60]
for k in range(n_pairs):
    p1 = int(starts[k]); p2 = p1 + int(isz[k]) - 150
    name = "r%08d" % k
    split = rng.random() < 0.02
    for mate, (p, mp, rev) in enumerate(((p1, p2, False), (p2, p1, True))):
        flag = 0x1 | 0x2 | (0x40 if mate == 0 else 0x80) | (0x10 if rev else 0x20)
        cigar = "150M" if not split else ("100M50S" if mate == 0 else "50S100M")
        tags = [("NM", "C", int(rng.integers(3))), ("RG", "Z", "rg")]
        i = int(rng.integers(4096))
        recs.append(dict(name=name, flag=flag, tid=0, pos=p, mapq=mqs[int(rng.integers(len(mqs)))], cigar=cigar, mtid=0, mpos=mp,
                         tlen=(int(isz[k]) if mate == 0 else -int(isz[k])), tags=tags,
                         seq4=seq_pool[i].tobytes(), qual=qual_pool[int(rng.integers(4096))].tobytes()))
recs.sort(key=lambda r: r["pos"])
if not os.path.exists(path):
    bw.write_bam(path, header, [("1", GENOME)], recs, block_bytes=65280)
print("BAM: %d records, %.1f MB, written in %.0f s" % (len(recs), os.path.getsize(path) / 1e6, time.time() - t0))
hist = {str(k): int(1000 * np.exp(-((k - 400) / 85.0) ** 2)) + 1 for k in range(160, 900)}
info = {"smp": {"mapped": len(recs), "unmapped": 0, "bam": path, "sample_name": "smp", "libraryArray": [
    {"library_name": "lib", "readgroups": ["rg"], "read_length": 150, "histogram": hist, "mean": 400.0, "sd": 60.0, "prevalence": 1.0}]}}
sites = []
for pos in range(20_000, GENOME - 20_000, SPACING):
    L = int(rng.integers(500, 3000))
    sites.append({"breakpoint": {"id": "d%d" % pos, "svtype": "DEL", "var_length": L,
                                 "A": {"chrom": "1", "pos": pos, "ci": [-10, 10], "is_reverse": False},
                                 "B": {"chrom": "1", "pos": pos + L + 1, "ci": [-10, 10], "is_reverse": True}}})
pybam = bam.AlignmentFile(path)
sample = library.Sample.from_lib_info(pybam, info, 1e-3)
for mode in (os.environ.get("SVT_INFLATE", "default"),):
    for th in THREADS:
        nb = nr.NativeBam(path)
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            off, frags, skipped = N._native_summaries(sites, sample, nb, nr.COUNT_CLASSIC, None, th)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print("inflate=%s threads=%2d: %d sites, %d fragments: %.3f s = %.0f sites/s (%.0f per thread)"
              % (mode, th, len(sites), len(frags), best, len(sites) / best, len(sites) / best / th))
