import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import goldenio as gio
from svtyper_amd import bam, geometry as geo, library, native_reads as nr, pipeline
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100
DATA = os.path.join(ROOT, "tests", "data")
BAM = os.path.join(DATA, "NA12878.target_loci.sorted.bam")
sites = gio.load("fixture_sites.json.gz")["sites"]
sample = library.Sample.from_lib_info(bam.AlignmentFile(BAM), json.load(open(os.path.join(DATA, "NA12878.bam.json"))), 1e-3)
nbam = nr.NativeBam(BAM)
tid_of = nbam.gettid
bps1 = np.concatenate([geo.breakpoint_record(s["breakpoint"], tid_of) for s in sites])
win1 = np.zeros(len(sites), nr.FETCH_DTYPE)
for k, s in enumerate(sites):
    bp = s["breakpoint"]
    for side, (t, lo, hi) in (("A", ("tid_a", "lo_a", "hi_a")), ("B", ("tid_b", "lo_b", "hi_b"))):
        chrom, a, b = pipeline.fetch_window(sample, bp[side]["chrom"], bp[side]["pos"], bp[side]["ci"], as_int=True)
        win1[t][k], win1[lo][k], win1[hi][k] = tid_of(chrom), int(a), int(b)
bps, win = np.tile(bps1, R), np.tile(win1, R)
rgs = list(sample.rg_to_lib.keys()); rg_lib = [0] * len(rgs)
for threads in [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]:
    t0 = time.perf_counter()
    off, frags, skipped = nbam.summarise(win, bps, rgs, rg_lib, 1000, nr.COUNT_SSO, threads)
    dt = time.perf_counter() - t0
    print("threads %3d: %d units, %d fragments in %.2f s = %.0f units/s" % (threads, len(win), len(frags), dt, len(win) / dt))
