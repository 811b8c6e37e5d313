"""Differential soak of the bulk VCF parser / emitter (include/svtyper_vcf.h) against the per-line Python model (svtyper_amd/vcf.py):
fixture lines with one to three field-level mutations stacked (tests/test_bulk_vcf.py::mutate) and, 40 % of the time, byte-level
damage on top.  Every line the parser TAKES must give Python's breakpoint dict and output text; a line Python crashes on must
never be taken.  CPU only.   python tests/soak_bulk_vcf.py <seed> <trials>   (profiles/r06_bulk_vcf_fuzz.txt)"""
import io
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_bulk_vcf as T
from svtyper_amd import bulk_vcf, hip
import numpy as np
head, body = T.fixture_lines()
bnd=[l for l in body if "SVTYPE=BND" in l]; plain=[l for l in body if "SVTYPE=BND" not in l]
vcf=T.header_vcf(head)
order=sorted(T.FIELDS, key=lambda k: vcf.format_rank[k]); fmt=":".join(order)
seed=int(sys.argv[1]); n=int(sys.argv[2])
rng=random.Random(seed)
taken=back=stopped=0
for trial in range(n):
    src=rng.choice(bnd) if rng.random()<0.15 else rng.choice(plain)
    line=src
    for _ in range(rng.choice([1,1,2,3])):      # several mutations stacked
        if len(line.rstrip("\n").split("\t"))<8: break
        try: line=T.mutate(line, rng)
        except (ValueError, IndexError): break
    if rng.random()<0.4:      # byte-level damage
        b=list(line.rstrip("\n"))
        for _ in range(rng.choice([1,1,2,4])):
            if not b: break
            i=rng.randrange(len(b)); op=rng.random()
            ch=rng.choice("\t\t;;==,,0123456789-+.eE[]<>:N ACGT\r\x0b\u00a0\u0663x_")
            if op<0.4: b[i]=ch
            elif op<0.7: b.insert(i,ch)
            else: del b[i]
        line="".join(b)+"\n"
    sum_quals=rng.random()<0.5; max_ci=rng.choice([1e10,50,0,3.5])
    parser=bulk_vcf.VcfParser(vcf,max_ci,sum_quals,False)
    chunk,used=parser.parse(line.encode("utf-8","surrogateescape"))
    model=T.header_vcf(head)
    old=sys.stderr; sys.stderr=io.StringIO()
    try:
        want=T.python_breakpoints(model,[line],max_ci,sum_quals); crashed=False
    except (SystemExit,Exception): want,crashed=None,True
    finally: sys.stderr=old
    if used==0: stopped+=1; continue
    kind=int(chunk.line_kind[0])
    if kind==bulk_vcf.LINE_PYTHON: back+=1; continue
    assert not crashed, repr(line)
    taken+=1
    if kind==bulk_vcf.LINE_HELD:
        assert want==[] and list(model._bnd_pending)==[line.split("\t")[2]], repr(line); continue
    T.assert_sites_equal(chunk, parser.chrom_names(), want)
    res=T.random_results(1,trial,skipped=False)
    text,_=chunk.emit(res,1,bulk_vcf.QUAL_SSO,order,False,fmt)
    first=want[0][3]
    if res.gt[0]>=0: first.qual+=float(res.sq[0])
    assert text.decode("utf-8","surrogateescape")==first.get_var_string_with(fmt,hip.format_results(res,order,False))+"\n", repr(line)
print("seed",seed,"trials",n,"taken",taken,"handed back",back,"stopped",stopped,"-- no difference")
