#!/usr/bin/env python
"""svtyper-sso command line end to end, one process vs N ranks under torch.distributed.run (on a one-GPU box the
ranks share the device and the 16-CPU quota, so this measures the host side of the sharded drivers: VCF parsing /
formatting in N interpreters, reader threads divided between them).  Checks the outputs are the same bytes."""
import json, os, pathlib, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_native_reads as N
import test_hip_geometry as G

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
RANKS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 4]
tmp = tempfile.mkdtemp()
bam = os.path.join(tmp, "s.bam")
_, info = N._synthetic_bam(bam, seed=100, n_pairs=900, sample="smp")
lib_json = os.path.join(tmp, "libs.json"); json.dump(info, open(lib_json, "w"))
_, vcf_path, _ = G._synthetic_case(pathlib.Path(tmp))
lines = open(vcf_path).read().splitlines(True)
hdr = [l for l in lines if l.startswith("#")]; body = [l for l in lines if not l.startswith("#") and "SVTYPE=BND" not in l]
big = os.path.join(tmp, "big.vcf"); open(big, "w").write("".join(hdr) + "".join(body * REP))
nv = len(body) * REP
env = dict(os.environ, PYTHONPATH=ROOT, SVT_TRACE=os.environ.get("SVT_TRACE", ""))
common = ["-m", "svtyper_amd.singlesample", "-i", big, "-B", bam, "-l", lib_json]

def run(cmd, out):
    t0 = time.perf_counter()
    subprocess.run(cmd + ["-o", out], check=True, env=env, cwd=ROOT)
    return time.perf_counter() - t0

t_import = run([sys.executable, "-m", "svtyper_amd.singlesample", "-i", vcf_path, "-B", bam, "-l", lib_json], os.path.join(tmp, "warm.vcf"))
single = os.path.join(tmp, "single.vcf")
t1 = run([sys.executable] + common, single)
print("%d variants | start-up (tiny input) %.2f s | 1 process %.2f s = %.0f variants/s" % (nv, t_import, t1, nv / t1))
want = open(single, "rb").read()
for n in RANKS:
    out = os.path.join(tmp, "r%d.vcf" % n)
    t = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
             "127.0.0.1", "--master-port", str(29500 + n)] + common, out)
    same = open(out, "rb").read() == want
    print("%d ranks: %.2f s = %.0f variants/s, output %s" % (n, t, nv / t, "identical" if same else "DIFFERS"))
