"""Fake alignments for geometry / tally tests (TEST INFRASTRUCTURE).

`FakeRead` quacks like the slice of pysam.AlignedSegment the SVTyper path touches (SURVEY.md
section 3.5) and can be rebuilt from a plain tuple, so golden fixtures can carry the reads
themselves.  `make_site` draws one synthetic breakpoint plus the reads around it: concordant and
discordant pairs, split reads with SA tags, soft-clipped reads, odd fragments (one primary, three
primaries, supplementary records), with coordinates jittered around the decision boundaries of
the reference's predicates.
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Tuple

_CONSUMES_REF = (True, False, True, True, False, False, False, True, True)
_ALIGNED = (True, False, False, False, False, False, False, True, True)
_OPS = "MIDNSHP=X"

READ_FIELDS = ("query_name", "flag", "reference_name", "reference_start", "cigarstring",
               "mapping_quality", "sa", "rg", "query_length", "template_length")


def parse_cigar(s: str) -> List[Tuple[int, int]]:
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((_OPS.index(ch), int(num)))
            num = ""
    return out


class FakeRead:
    def __init__(self, query_name, flag, reference_name, reference_start, cigarstring, mapping_quality,
                 sa=None, rg="rg0", query_length=0, template_length=0):
        self.query_name = query_name
        self.flag = int(flag)
        self.reference_name = reference_name
        self.reference_start = int(reference_start)
        self.cigarstring = cigarstring
        self.cigar = parse_cigar(cigarstring)
        self.mapping_quality = int(mapping_quality)
        self.query_length = int(query_length)
        self.template_length = int(template_length)
        self._tags: Dict[str, object] = {"RG": rg}
        if sa:
            self._tags["SA"] = sa

    def astuple(self):
        return (self.query_name, self.flag, self.reference_name, self.reference_start, self.cigarstring,
                self.mapping_quality, self._tags.get("SA"), self._tags["RG"], self.query_length,
                self.template_length)

    # flag bits
    is_unmapped = property(lambda s: bool(s.flag & 0x4))
    mate_is_unmapped = property(lambda s: bool(s.flag & 0x8))
    is_reverse = property(lambda s: bool(s.flag & 0x10))
    mate_is_reverse = property(lambda s: bool(s.flag & 0x20))
    is_secondary = property(lambda s: bool(s.flag & 0x100))
    is_duplicate = property(lambda s: bool(s.flag & 0x400))
    is_supplementary = property(lambda s: bool(s.flag & 0x800))

    @property
    def pos(self):
        return self.reference_start

    @property
    def cigartuples(self):
        return self.cigar

    @property
    def reference_end(self):
        return self.reference_start + sum(n for op, n in self.cigar if _CONSUMES_REF[op])

    @property
    def query_alignment_length(self):
        return sum(n for op, n in self.cigar if op in (0, 1, 7, 8))

    def infer_query_length(self):
        return sum(n for op, n in self.cigar if op in (0, 1, 4, 7, 8))

    def get_overlap(self, start, end):
        ov, p = 0, self.reference_start
        for op, n in self.cigar:
            if _ALIGNED[op]:
                lo, hi = max(p, start), min(p + n, end)
                if hi > lo:
                    ov += hi - lo
            if _CONSUMES_REF[op]:
                p += n
        return ov

    def has_tag(self, k):
        return k in self._tags

    def get_tag(self, k):
        return self._tags[k]

    def set_tag(self, k, v, value_type=None):
        self._tags[k] = v


# --------------------------------------------------------------------------- synthetic sites
def _mapq(rng: random.Random) -> int:
    u = rng.random()
    if u < 0.70:
        return 60
    if u < 0.78:
        return 0
    if u < 0.86:
        return rng.choice([3, 10, 20, 30, 40])
    if u < 0.90:
        return 255
    return rng.randint(1, 59)


def make_libraries(rng: random.Random, n_libs: int):
    """[(name, readgroups, mean, sd, read_length, hist dict)]"""
    libs = []
    for i in range(n_libs):
        mu = rng.choice([300, 320, 400, 450])
        sd = rng.choice([30, 50, 80])
        hist: Dict[int, int] = {}
        for _ in range(4000):
            x = int(round(rng.gauss(mu, sd)))
            if x >= 1:
                hist[x] = hist.get(x, 0) + 1
        tot = float(sum(hist.values()))
        mean = sum(k * v for k, v in hist.items()) / tot
        var = sum(v * (k - mean) ** 2 for k, v in hist.items()) / tot
        sdev = var ** 0.5
        if rng.random() < 0.3:  # make mean + 3 sd integral for some libraries (Counter float-key path)
            mean = float(round(mean))
            sdev = float(round(sdev))
        libs.append(("lib%d" % i, ["rg%d" % i], mean, sdev, 101, hist))
    return libs


def make_site(rng: random.Random, site_id: str, libs) -> Tuple[dict, List[FakeRead]]:
    """One breakpoint dict (as svtyper/parsers.py:149-154,190-203 builds it) and its reads, in the
    order a BAM fetch of region A then region B would deliver them."""
    svtype = rng.choice(["DEL", "DEL", "DEL", "DUP", "INV", "BND"])
    chrom = rng.choice(["1", "2"])
    posA = rng.randint(20000, 200000)
    length = int(rng.choice([30, 80, 150, 300, 600, 1200, 5000, 40000]) * rng.uniform(0.8, 1.2))
    posB = posA + length
    chromB = chrom
    ci = lambda: rng.choice([[0, 0], [-10, 10], [-3, 5], [-50, 50]])
    bp = {"id": site_id, "svtype": svtype,
          "A": {"chrom": chrom, "pos": posA, "ci": ci(), "is_reverse": False},
          "B": {"chrom": chromB, "pos": posB, "ci": ci(), "is_reverse": True}}
    if svtype == "DEL":
        bp["var_length"] = posB - posA
        o1, o2 = False, True
    elif svtype == "DUP":
        o1, o2 = True, False
    elif svtype == "INV":
        o1, o2 = False, False
    else:
        o1, o2 = rng.choice([(False, True), (True, False), (False, False), (True, True)])
        if rng.random() < 0.5:
            bp["B"]["chrom"] = chromB = "2" if chrom == "1" else "1"
            bp["B"]["pos"] = posB = rng.randint(20000, 200000)
    bp["A"]["is_reverse"], bp["B"]["is_reverse"] = o1, o2
    if o1:
        bp["A"]["pos"] += 1
    if o2:
        bp["B"]["pos"] += 1
    posA, posB = bp["A"]["pos"], bp["B"]["pos"]

    reads: List[FakeRead] = []
    n_frag = rng.randint(4, 45)
    for k in range(n_frag):
        name = "%s.f%03d" % (site_id, rng.randint(0, 999))
        lib = rng.choice(libs)
        rg = lib[1][0]
        mean, sd = lib[2], lib[3]
        isize = max(120, int(rng.gauss(mean, sd)))
        kind = rng.random()
        rl = 101
        jitter = lambda w=25: rng.randint(-w, w)
        if kind < 0.30:
            # concordant pair straddling side A or B (reference support)
            p = (posA if rng.random() < 0.5 else posB) if chromB == chrom else posA
            c = chrom
            if chromB != chrom and rng.random() < 0.5:
                p, c = posB, chromB
            s = p - rng.randint(0, isize) + jitter(40)
            a = FakeRead(name, 99, c, s, "%dM" % rl, _mapq(rng), rg=rg, template_length=isize)
            b = FakeRead(name, 147, c, s + isize - rl, "%dM" % rl, _mapq(rng), rg=rg, template_length=-isize)
            reads += [a, b]
        elif kind < 0.55:
            # alt-supporting pair: orientation per breakpoint strands, ends around the two breakends
            sa_ = posA - rng.randint(20, int(mean)) + jitter() if not o1 else posA + rng.randint(0, int(mean) - rl) + jitter()
            sb_ = posB + rng.randint(0, int(mean) - rl) + jitter() if o2 else posB - rng.randint(rl, int(mean)) + jitter()
            fa = 65 | (0x10 if o1 else 0) | (0x20 if o2 else 0)
            fb = 129 | (0x10 if o2 else 0) | (0x20 if o1 else 0)
            a = FakeRead(name, fa, chrom, max(1, sa_), "%dM" % rl, _mapq(rng), rg=rg)
            b = FakeRead(name, fb, chromB, max(1, sb_), "%dM" % rl, _mapq(rng), rg=rg)
            if rng.random() < 0.2 and svtype == "INV":  # reciprocal orientation
                a.flag ^= 0x10
                b.flag ^= 0x10
            reads += [a, b]
        elif kind < 0.72:
            # split read: primary piece ends at A (+- slop), supplementary starts at B
            off = rng.choice([0, 0, 1, -2, 3, -3, 4, -5])
            m = rng.randint(30, 70)
            if not o1:
                start = posA + off - m
                cig = "%dM%dS" % (m, rl - m)
            else:
                start = posA + off
                cig = "%dS%dM" % (rl - m, m)
            off2 = rng.choice([0, 0, 1, -1, 3, -4])
            m2 = rl - m + rng.choice([0, 0, 5, -5])
            m2 = min(max(m2, 10), rl - 5)
            if o2:
                sa_pos = posB + off2
                sa_cig = "%dS%dM" % (rl - m2, m2)
            else:
                sa_pos = posB + off2 - m2
                sa_cig = "%dM%dS" % (m2, rl - m2)
            strand_a = rng.random() < 0.5
            strand_b = strand_a if svtype != "INV" else not strand_a
            sa = "%s,%d,%s,%s,%d,0;" % (chromB, sa_pos + 1, "-" if strand_b else "+", sa_cig, _mapq(rng))
            if rng.random() < 0.08:
                sa += "%s,%d,+,50M51S,30,0;" % (chrom, posA + 5000)  # two SA entries -> invalid candidate
            a = FakeRead(name, 65 | (0x10 if strand_a else 0), chrom, max(1, start), cig, _mapq(rng), sa=sa, rg=rg)
            reads.append(a)
            if rng.random() < 0.7:  # its mate
                b = FakeRead(name, 129 | (0x10 if not strand_a else 0), chrom,
                             max(1, start + rng.randint(-400, 400)), "%dM" % rl, _mapq(rng), rg=rg)
                if rng.random() < 0.25:
                    b.cigar = parse_cigar("71M30S"); b.cigarstring = "71M30S"  # soft clip on the mate too
                reads.append(b)
            if rng.random() < 0.3:  # supplementary record of the same name: not a primary
                reads.append(FakeRead(name, 2113, chromB, max(1, sa_pos), sa_cig.replace("S", "H"), 60, rg=rg))
        elif kind < 0.88:
            # soft-clipped read without SA, clip at the breakpoint (+- slop)
            off = rng.choice([0, 0, 2, -3, 4, -4])
            m = rng.randint(45, 85)
            side_b = rng.random() < 0.5
            if not side_b:
                p, c, rev = posA, chrom, o1
            else:
                p, c, rev = posB, chromB, o2
            if not rev:
                start, cig = p + off - m, "%dM%dS" % (m, rl - m)
            else:
                start, cig = p + off, "%dS%dM" % (rl - m, m)
            if rng.random() < 0.15:
                cig = "10S" + cig if cig[0].isdigit() and "S" not in cig.split("M")[0] else cig
            a = FakeRead(name, 73 if rng.random() < 0.4 else 65, c, max(1, start), cig, _mapq(rng), rg=rg,
                         query_length=rng.choice([0, 0, rl, 160]))
            reads.append(a)
            if a.flag == 65:
                reads.append(FakeRead(name, 145, c, max(1, start + rng.randint(100, 500)), "%dM" % rl,
                                      _mapq(rng), rg=rg))
        elif kind < 0.94:
            # reads covering the breakpoint with indels in the CIGAR (get_overlap / reference_end)
            p = posA if rng.random() < 0.5 else posB
            c = chrom if p == posA else chromB
            cig = rng.choice(["50M2D51M", "30M1I70M", "40M100N61M", "101M", "20M5D30M3I48M"])
            s = p - rng.randint(10, 90)
            reads.append(FakeRead(name, 99, c, max(1, s), cig, _mapq(rng), rg=rg, template_length=isize))
            reads.append(FakeRead(name, 147, c, max(1, s + isize - rl), "%dM" % rl, _mapq(rng), rg=rg))
        else:
            # odd fragments: three primaries with one name, or the same record twice
            s = posA - rng.randint(50, 300)
            r1 = FakeRead(name, 99, chrom, max(1, s), "%dM" % rl, _mapq(rng), rg=rg)
            r2 = FakeRead(name, 147, chrom, max(1, s + isize - rl), "%dM" % rl, _mapq(rng), rg=rg)
            reads += [r1, r2]
            if rng.random() < 0.5:
                reads.append(FakeRead(name, 65, chrom, max(1, posA - 60), "60M41S", _mapq(rng), rg=rg))
            else:
                reads.append(FakeRead(name, 99, chrom, max(1, s), "%dM" % rl, r1.mapping_quality, rg=rg))
    # fetch order: region A then region B, each by coordinate (duplicates across regions allowed)
    reads.sort(key=lambda r: (0 if r.reference_name == chrom else 1, r.reference_start))
    return bp, reads
