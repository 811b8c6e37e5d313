"""Read-fragment model and the integer geometry questions the evidence packer asks.

Host-side restatement (written for this package, Python 3) of the reference's fragment layer:
svtyper/parsers.py:729-857 (SamFragment) and :891-1253 (SplitRead / SplitPiece / QueryPos).
It answers, per read-fragment and breakpoint, the yes/no questions whose answers become the
gated MAPQ bytes and straddle bits of an evidence record (packer.py); the weighting, the
insert-size test and all sums happen on the device.

Only pysam-style *attributes* of the reads are used, so any object with
those attributes works: svtyper_amd.bam.AlignedSegment, pysam.AlignedSegment, tests' FakeRead.
"""
from __future__ import annotations

import re
from typing import List, Optional, Sequence, Tuple

CLIP_OPS = (4, 5)            # S, H
QUERY_OPS = (0, 1, 7, 8)     # M, I, =, X consume the query
REF_OPS = (0, 2, 3, 7, 8)    # M, D, N, =, X consume the reference
_CIGAR_CODE = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}
_CIGAR_RE = re.compile(r"(\d+)([MIDNSHPX=])")


def cigarstring_to_tuple(cigarstring: str) -> List[Tuple[int, int]]:
    """'5H3S2D' -> [(5, 5), (4, 3), (2, 2)]   (parsers.py:1080-1086)"""
    return [(_CIGAR_CODE[op], int(n)) for n, op in _CIGAR_RE.findall(cigarstring)]


def reference_end_from_cigar(reference_start: int, cigar: Sequence[Tuple[int, int]]) -> int:
    """Coordinate just past the last aligned base (parsers.py:1088-1101)."""
    return reference_start + sum(n for op, n in cigar if op in REF_OPS)


class QueryPos:
    """Aligned interval of the query, in read orientation (parsers.py:1257-1264)."""
    __slots__ = ("query_start", "query_end", "query_length")

    def __init__(self, query_start, query_end, query_length):
        self.query_start = int(query_start)
        self.query_end = int(query_end)
        self.query_length = int(query_length)


def query_pos_from_cigar(cigar: Sequence[Tuple[int, int]], is_reverse: bool) -> QueryPos:
    """parsers.py:922-947.  Only a clip that is the FIRST operation (in read orientation) shifts
    the start; later clips just extend the query length."""
    ops = list(cigar)[::-1] if is_reverse else list(cigar)
    start = end = length = 0
    for i, (op, n) in enumerate(ops):
        if op in CLIP_OPS:
            if i == 0:
                start += n
                end += n
            length += n
        elif op in QUERY_OPS:
            end += n
            length += n
    return QueryPos(start, end, length)


class SplitPiece:
    """One alignment of a chimeric read (parsers.py:902-950)."""
    __slots__ = ("chrom", "reference_start", "reference_end", "is_reverse", "cigar", "mapping_quality",
                 "query_pos")

    def __init__(self, chrom, reference_start, is_reverse, cigar, mapq, reference_end=None):
        self.chrom = chrom
        self.reference_start = reference_start
        self.reference_end = reference_end
        self.is_reverse = is_reverse
        self.cigar = cigar
        self.mapping_quality = mapq
        self.query_pos = query_pos_from_cigar(cigar, is_reverse)

    def set_reference_end(self, reference_end):
        self.reference_end = reference_end

    # static spellings kept for API parity with the reference's tests (tests/test_svtyper.py:19-34)
    get_query_pos_from_cigar = staticmethod(query_pos_from_cigar)


def _is_clip(op: int) -> bool:
    return op == 4 or op == 5


def _left_clipped(cigar) -> bool:
    """Is the longest clip on the reference-left end?  (parsers.py:1242-1253)"""
    (lop, llen), (rop, rlen) = cigar[0], cigar[-1]
    lc, rc = _is_clip(lop), _is_clip(rop)
    return (lc and not rc) or (lc and rc and llen > rlen)


def start_diagonal(piece: SplitPiece) -> int:
    """Reference position where the alignment would start had the whole query aligned
    (parsers.py:1062-1067)."""
    qp = piece.query_pos
    lead = qp.query_length - qp.query_end if piece.is_reverse else qp.query_start
    return piece.reference_start - lead


def end_diagonal(piece: SplitPiece) -> int:
    """parsers.py:1071-1076"""
    qp = piece.query_pos
    aligned = qp.query_length - qp.query_start if piece.is_reverse else qp.query_end
    return piece.reference_end - aligned


def supports_breakend(piece: SplitPiece, chrom, pos, is_reverse, slop) -> bool:
    """Does the piece end (forward side) / start (reverse side) within `slop` of the breakend?
    (parsers.py:1121-1134)"""
    if piece.chrom != chrom:
        return False
    coord = piece.reference_start if is_reverse else piece.reference_end
    return pos - slop <= coord <= pos + slop


class SplitRead:
    """A primary read seen as a two-piece chimeric alignment (parsers.py:891-1253)."""

    SplitPiece = SplitPiece
    cigarstring_to_tuple = staticmethod(cigarstring_to_tuple)
    get_reference_end_from_cigar = staticmethod(reference_end_from_cigar)
    get_start_diagonal = staticmethod(start_diagonal)
    get_end_diagonal = staticmethod(end_diagonal)
    check_split_support = staticmethod(supports_breakend)

    def __init__(self, read, lib):
        self.query_name = read.query_name
        self.read = read
        self.lib = lib
        self.sa = None
        self.is_soft_clip = False
        self.query_left: Optional[SplitPiece] = None
        self.query_right: Optional[SplitPiece] = None

    def _order_by_clip(self, a: SplitPiece, b: SplitPiece):
        if _left_clipped(a.cigar):
            self.query_left, self.query_right = b, a
        else:
            self.query_left, self.query_right = a, b

    def _primary_piece(self) -> SplitPiece:
        r = self.read
        return SplitPiece(r.reference_name, r.reference_start, r.is_reverse, r.cigar, r.mapping_quality,
                          r.reference_end)

    def non_overlap(self) -> int:
        """Smaller count of query bases private to one piece (parsers.py:1103-1119)."""
        l, r = self.query_left.query_pos, self.query_right.query_pos
        shared = max(0, 1 + min(l.query_end, r.query_end) - max(l.query_start, r.query_start))
        return min(1 + l.query_end - l.query_start - shared, 1 + r.query_end - r.query_start - shared)

    def is_valid(self, min_non_overlap=20, min_indel=50, max_unmapped_bases=50) -> bool:
        """QC + population of query_left/query_right (parsers.py:959-1058)."""
        read = self.read
        if not read.has_tag("SA"):
            # soft-clipped read without a split alignment: counts as a "clip" candidate whose other
            # piece is a dummy with MAPQ 0 (parsers.py:964-988)
            cig = read.cigar
            first_clip, last_clip = _is_clip(cig[0][0]), _is_clip(cig[-1][0])
            if not (first_clip or last_clip):
                return False
            clip_length = max(cig[0][1] * first_clip, cig[-1][1] * last_clip)
            if clip_length > 0 and (read.query_length - read.query_alignment_length) <= max_unmapped_bases:
                dummy = SplitPiece(None, 1, read.is_reverse, cig, 0, 1)
                self._order_by_clip(self._primary_piece(), dummy)
                self.is_soft_clip = True
                return True
            return False

        entries = read.get_tag("SA").rstrip(";").split(";")
        if len(entries) > 1:          # more than two pieces: discarded (parsers.py:992-993)
            return False
        self.sa = entries[0].split(",")
        mate_chrom = self.sa[0]
        mate_pos = int(self.sa[1]) - 1            # SA is 1-based
        mate_cigar = cigarstring_to_tuple(self.sa[3])
        a = self._primary_piece()
        b = SplitPiece(mate_chrom, mate_pos, self.sa[2] == "-", mate_cigar, int(self.sa[4]),
                       reference_end_from_cigar(mate_pos, mate_cigar))
        if read.reference_name == mate_chrom:     # left/right by reference position (:1020-1028)
            if read.reference_start > mate_pos:
                self.query_left, self.query_right = b, a
            else:
                self.query_left, self.query_right = a, b
        else:
            self._order_by_clip(a, b)

        if self.non_overlap() < min_non_overlap:
            return False

        left, right = self.query_left, self.query_right
        if left.chrom == right.chrom and left.is_reverse == right.is_reverse:
            # off-diagonal distance and unaligned "desert" between the pieces (:1036-1055)
            if left.is_reverse:
                ins_size = end_diagonal(right) - start_diagonal(left)
            else:
                ins_size = end_diagonal(left) - start_diagonal(right)
            if abs(ins_size) < min_indel:
                return False
            desert = right.query_pos.query_start - left.query_pos.query_end - 1
            if desert > 0 and desert - max(0, ins_size) > max_unmapped_bases:
                return False
        return True

    def is_split_straddle(self, chromA, posA, ciA, chromB, posB, ciB, o1_is_reverse, o2_is_reverse,
                          svtype, split_slop) -> Tuple[bool, bool]:
        """(left piece supports, right piece supports)   (parsers.py:1136-1215)"""
        if chromA != chromB or posA > posB:
            lo = (chromB, posB, o2_is_reverse)
            hi = (chromA, posA, o1_is_reverse)
        else:
            lo = (chromA, posA, o1_is_reverse)
            hi = (chromB, posB, o2_is_reverse)
        L, R = self.query_left, self.query_right
        hit = lambda piece, bp: supports_breakend(piece, bp[0], bp[1], bp[2], split_slop)
        if (not self.is_soft_clip) or svtype == "DEL" or svtype == "INS":
            return hit(L, lo), hit(R, hi)
        if svtype == "DUP":
            return hit(L, hi), hit(R, lo)
        if svtype == "INV":
            return (hit(L, lo) or hit(L, hi)), (hit(R, lo) or hit(R, hi))
        return False, False


class SamFragment:
    """All alignments of one molecule (query name) around a breakpoint (parsers.py:729-857)."""

    def __init__(self, read, lib):
        self.lib = lib
        self.query_name = read.query_name
        self.primary_reads: list = []
        self.split_reads: List[SplitRead] = []
        self.num_primary = 0
        self.readA = None
        self.readB = None
        self._seen = set()
        self.add_read(read)

    def add_read(self, read):
        key = (read.query_name, read.flag)         # one record per (name, flag) (parsers.py:726-754)
        if key in self._seen:
            return
        self._seen.add(key)
        if read.is_secondary or read.is_supplementary:
            return
        self.primary_reads.append(read)
        self.num_primary += 1
        # a mapped read without a CIGAR ('*') cannot be a split candidate (the reference indexes cigar[0] and
        # dies on such a record; one odd read must not end a whole-genome job)
        if read.cigar:
            candidate = SplitRead(read, self.lib)
            if candidate.is_valid():
                self.split_reads.append(candidate)
        if self.num_primary == 2:
            self.readA, self.readB = self.primary_reads

    def get_ispan(self, min_aligned):
        return (self.readA.reference_start + min_aligned, self.readB.reference_end - min_aligned - 1)

    def get_ospan(self):
        return (self.readA.reference_start, self.readB.reference_end)

    def is_ref_seq(self, read, variant, chrom, pos, ci, min_aligned) -> bool:
        """min_aligned reference-matching bases on both sides of pos (parsers.py:801-816)."""
        if read.reference_name != chrom:
            return False
        return read.get_overlap(max(0, pos - min_aligned), pos + min_aligned) >= 2 * min_aligned

    def is_pair_straddle(self, chromA, posA, ciA, chromB, posB, ciB, o1_is_reverse, o2_is_reverse,
                         min_aligned, lib) -> bool:
        """Orientation, chromosome and inner-span test of the primary pair (parsers.py:821-857)."""
        if self.num_primary != 2:
            return False
        a, b = self.readA, self.readB
        if a.is_reverse != o1_is_reverse or b.is_reverse != o2_is_reverse:
            return False
        if a.reference_name != chromA or b.reference_name != chromB:
            return False
        i1, i2 = self.get_ispan(min_aligned)
        flank = lib.mean + lib.sd * 3
        for inner, pos, ci, rev in ((i1, posA, ciA, o1_is_reverse), (i2, posB, ciB, o2_is_reverse)):
            if rev:
                if inner < pos + ci[0] or inner > pos + ci[1] + flank:
                    return False
            else:
                if inner > pos + ci[1] or inner < pos + ci[0] - flank:
                    return False
        return True
