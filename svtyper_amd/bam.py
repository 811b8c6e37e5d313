"""Minimal BGZF / BAM / BAI reader with the slice of the pysam API the SVTyper path relies on.

pysam (htslib) is what the reference uses (svtyper/classic.py:126-129, parsers.py:480-558); it
is not available in this image and cannot be installed, so the host side ships its own reader.
Only what SURVEY.md section 3.5 lists is implemented:

    AlignmentFile(path, 'rb'): filename, header['RG'], references, lengths, gettid(), fetch(),
                               count(read_callback='all'), mapped, unmapped, close()
    AlignedSegment: query_name, flag and its bits, reference_id/name, reference_start (= pos),
                    reference_end, mapping_quality, template_length, cigar (= cigartuples),
                    has_tag/get_tag/set_tag, get_overlap, query_length, query_alignment_length,
                    infer_query_length

CRAM is not supported (the reference needs htslib + a reference FASTA for it).
If a real pysam is importable, `open_alignment_file` prefers it.
"""
from __future__ import annotations

import os
import struct
import zlib
from collections import OrderedDict
from typing import Dict, Iterator, List, Optional, Tuple

# CIGAR op codes: M I D N S H P = X
_CONSUMES_REF = (True, False, True, True, False, False, False, True, True)
_ALIGNED = (True, False, False, False, False, False, False, True, True)  # M = X


class BgzfReader:
    """Random access over a BGZF file through (compressed offset, in-block offset) addresses."""

    def __init__(self, path: str, cache_blocks: int = 64):
        self._f = open(path, "rb")
        self._cache: "OrderedDict[int, Tuple[bytes, int]]" = OrderedDict()
        self._cache_blocks = cache_blocks
        self._coff = 0   # compressed offset of the current block
        self._uoff = 0   # offset inside the current block
        self._block = b""
        self._next = 0   # compressed offset of the next block

    def close(self):
        self._f.close()

    def _load(self, coff: int) -> bool:
        hit = self._cache.get(coff)
        if hit is not None:
            self._cache.move_to_end(coff)
            self._block, self._next = hit
            self._coff = coff
            return len(self._block) > 0 or self._next > coff
        self._f.seek(coff)
        hdr = self._f.read(18)
        if len(hdr) < 18:
            self._block, self._next, self._coff = b"", coff, coff
            return False
        if hdr[0] != 31 or hdr[1] != 139:
            raise IOError("not a BGZF block at offset %d" % coff)
        xlen = struct.unpack_from("<H", hdr, 10)[0]
        extra = hdr[12:18] + self._f.read(xlen - 6)
        bsize = None
        i = 0
        while i + 4 <= len(extra):
            si1, si2, slen = extra[i], extra[i + 1], struct.unpack_from("<H", extra, i + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", extra, i + 4)[0]
            i += 4 + slen
        if bsize is None:
            raise IOError("BGZF block without BC field")
        cdata_len = bsize - xlen - 19
        cdata = self._f.read(cdata_len)
        self._f.read(8)  # crc32 + isize
        data = zlib.decompress(cdata, -15) if cdata_len > 0 else b""
        nxt = coff + bsize + 1
        self._cache[coff] = (data, nxt)
        if len(self._cache) > self._cache_blocks:
            self._cache.popitem(last=False)
        self._block, self._next, self._coff = data, nxt, coff
        return True

    def seek(self, voffset: int):
        coff, uoff = voffset >> 16, voffset & 0xFFFF
        self._load(coff)
        self._uoff = uoff

    def tell(self) -> int:
        if self._uoff >= len(self._block) and self._block:
            return self._next << 16
        return (self._coff << 16) | self._uoff

    def read(self, n: int) -> bytes:
        out = []
        need = n
        while need > 0:
            avail = len(self._block) - self._uoff
            if avail <= 0:
                if not self._load(self._next):
                    break
                self._uoff = 0
                if not self._block:
                    # empty block (EOF marker) -- try the next one
                    if self._next == self._coff:
                        break
                    continue
                continue
            take = min(avail, need)
            out.append(self._block[self._uoff:self._uoff + take])
            self._uoff += take
            need -= take
        return b"".join(out)


class AlignedSegment:
    """One BAM record; attribute names follow pysam."""

    __slots__ = ("_file", "query_name", "flag", "reference_id", "reference_start", "mapping_quality",
                 "next_reference_id", "next_reference_start", "template_length", "cigar", "query_length",
                 "_tagbytes", "_tags", "_ref_end", "_raw_seq")

    def __init__(self, afile, data: bytes):
        (ref_id, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, next_ref, next_pos, tlen) = \
            struct.unpack_from("<iiBBHHHiiii", data, 0)
        self._file = afile
        self.reference_id = ref_id
        self.reference_start = pos
        self.mapping_quality = mapq
        self.flag = flag
        self.next_reference_id = next_ref
        self.next_reference_start = next_pos
        self.template_length = tlen
        off = 32
        self.query_name = data[off:off + l_read_name - 1].decode("ascii")
        off += l_read_name
        if n_cigar:
            raw = struct.unpack_from("<%dI" % n_cigar, data, off)
            self.cigar = [(c & 0xF, c >> 4) for c in raw]
        else:
            self.cigar = []
        off += 4 * n_cigar
        self.query_length = l_seq                      # pysam: l_seq of the record
        self._raw_seq = None
        off += (l_seq + 1) // 2 + l_seq
        self._tagbytes = data[off:]
        self._tags: Optional[Dict[str, object]] = None
        self._ref_end = None

    # ---- flag bits
    @property
    def is_paired(self): return bool(self.flag & 0x1)
    @property
    def is_proper_pair(self): return bool(self.flag & 0x2)
    @property
    def is_unmapped(self): return bool(self.flag & 0x4)
    @property
    def mate_is_unmapped(self): return bool(self.flag & 0x8)
    @property
    def is_reverse(self): return bool(self.flag & 0x10)
    @property
    def mate_is_reverse(self): return bool(self.flag & 0x20)
    @property
    def is_read1(self): return bool(self.flag & 0x40)
    @property
    def is_read2(self): return bool(self.flag & 0x80)
    @property
    def is_secondary(self): return bool(self.flag & 0x100)
    @property
    def is_qcfail(self): return bool(self.flag & 0x200)
    @property
    def is_duplicate(self): return bool(self.flag & 0x400)
    @property
    def is_supplementary(self): return bool(self.flag & 0x800)

    # ---- coordinates
    @property
    def pos(self):
        return self.reference_start

    @property
    def reference_name(self):
        return self._file.references[self.reference_id] if self.reference_id >= 0 else None

    @property
    def reference_end(self):
        """pos + sum of the reference-consuming CIGAR operations (None when there is no CIGAR)."""
        if self._ref_end is None:
            if not self.cigar:
                return None
            end = self.reference_start
            for op, n in self.cigar:
                if _CONSUMES_REF[op]:
                    end += n
            self._ref_end = end
        return self._ref_end

    @property
    def cigartuples(self):
        return self.cigar

    @property
    def query_alignment_length(self):
        return sum(n for op, n in self.cigar if op in (0, 1, 7, 8))

    def infer_query_length(self):
        return sum(n for op, n in self.cigar if op in (0, 1, 4, 7, 8))

    def get_overlap(self, start: int, end: int) -> int:
        """Number of M/=/X-aligned reference bases inside [start, end)."""
        ov = 0
        p = self.reference_start
        for op, n in self.cigar:
            if _ALIGNED[op]:
                lo = p if p > start else start
                hi = p + n if p + n < end else end
                if hi > lo:
                    ov += hi - lo
            if _CONSUMES_REF[op]:
                p += n
        return ov

    # ---- tags
    def _parse_tags(self):
        tags: Dict[str, object] = {}
        b = self._tagbytes
        i, n = 0, len(b)
        while i + 3 <= n:
            key = b[i:i + 2].decode("ascii")
            t = chr(b[i + 2])
            i += 3
            if t == "A":
                val = chr(b[i]); i += 1
            elif t == "c":
                val = struct.unpack_from("<b", b, i)[0]; i += 1
            elif t == "C":
                val = b[i]; i += 1
            elif t == "s":
                val = struct.unpack_from("<h", b, i)[0]; i += 2
            elif t == "S":
                val = struct.unpack_from("<H", b, i)[0]; i += 2
            elif t == "i":
                val = struct.unpack_from("<i", b, i)[0]; i += 4
            elif t == "I":
                val = struct.unpack_from("<I", b, i)[0]; i += 4
            elif t == "f":
                val = struct.unpack_from("<f", b, i)[0]; i += 4
            elif t in "ZH":
                j = b.index(b"\0", i)
                val = b[i:j].decode("ascii"); i = j + 1
            elif t == "B":
                sub = chr(b[i]); cnt = struct.unpack_from("<I", b, i + 1)[0]; i += 5
                fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
                sz = struct.calcsize(fmt)
                val = list(struct.unpack_from("<%d%s" % (cnt, fmt), b, i)); i += cnt * sz
            else:
                raise ValueError("unknown BAM tag type %r" % t)
            tags[key] = val
        self._tags = tags

    def has_tag(self, key: str) -> bool:
        if self._tags is None:
            self._parse_tags()
        return key in self._tags

    def get_tag(self, key: str):
        if self._tags is None:
            self._parse_tags()
        return self._tags[key]  # KeyError like pysam

    def set_tag(self, key: str, value, value_type=None):
        if self._tags is None:
            self._parse_tags()
        self._tags[key] = value

    def __repr__(self):
        return "<AlignedSegment %s flag=%d %s:%d mapq=%d>" % (
            self.query_name, self.flag, self.reference_name, self.reference_start, self.mapping_quality)


def _reg2bins(beg: int, end: int) -> List[int]:
    """UCSC binning scheme bins overlapping [beg, end) (SAM spec 5.3)."""
    end -= 1
    bins = [0]
    for shift, off in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins.extend(range(off + (beg >> shift), off + (end >> shift) + 1))
    return bins


class AlignmentFile:
    """BAM file opened for reading, with its .bai index when present."""

    def __init__(self, path: str, mode: str = "rb", **kwargs):
        if "c" in mode or path.endswith(".cram"):
            raise NotImplementedError("CRAM needs htslib; this reader handles BAM only")
        self.filename = path
        self._bgzf = BgzfReader(path)
        if self._bgzf.read(4) != b"BAM\1":
            raise IOError("%s is not a BAM file" % path)
        l_text = struct.unpack("<i", self._bgzf.read(4))[0]
        self.text = self._bgzf.read(l_text).split(b"\0", 1)[0].decode("ascii", "replace")
        n_ref = struct.unpack("<i", self._bgzf.read(4))[0]
        refs, lens = [], []
        for _ in range(n_ref):
            l_name = struct.unpack("<i", self._bgzf.read(4))[0]
            refs.append(self._bgzf.read(l_name)[:-1].decode("ascii"))
            lens.append(struct.unpack("<i", self._bgzf.read(4))[0])
        self.references = tuple(refs)
        self.lengths = tuple(lens)
        self._tid = {r: i for i, r in enumerate(refs)}
        self._first_record = self._bgzf.tell()
        self.header = self._parse_header(self.text)
        self._index = None
        self._index_stats = None
        for cand in (path + ".bai", os.path.splitext(path)[0] + ".bai"):
            if os.path.exists(cand):
                self._load_index(cand)
                break

    @staticmethod
    def _parse_header(text: str) -> Dict[str, list]:
        hdr: Dict[str, list] = {}
        for line in text.splitlines():
            if not line.startswith("@") or line.startswith("@CO"):
                continue
            parts = line.split("\t")
            rec = {}
            for f in parts[1:]:
                if len(f) >= 3 and f[2] == ":":
                    rec[f[:2]] = f[3:]
            hdr.setdefault(parts[0][1:], []).append(rec)
        return hdr

    def _load_index(self, path: str):
        with open(path, "rb") as f:
            data = f.read()
        if data[:4] != b"BAI\1":
            raise IOError("%s is not a BAI index" % path)
        off = 4
        n_ref = struct.unpack_from("<i", data, off)[0]; off += 4
        index = []
        mapped = unmapped = 0
        for _ in range(n_ref):
            n_bin = struct.unpack_from("<i", data, off)[0]; off += 4
            bins: Dict[int, List[Tuple[int, int]]] = {}
            for _ in range(n_bin):
                b, n_chunk = struct.unpack_from("<Ii", data, off); off += 8
                chunks = list(struct.iter_unpack("<QQ", data[off:off + 16 * n_chunk])); off += 16 * n_chunk
                if b == 37450:  # pseudo-bin: [unmapped beg/end], [n_mapped, n_unmapped]
                    if len(chunks) >= 2:
                        mapped += chunks[1][0]
                        unmapped += chunks[1][1]
                else:
                    bins[b] = chunks
            n_intv = struct.unpack_from("<i", data, off)[0]; off += 4
            linear = list(struct.unpack_from("<%dQ" % n_intv, data, off)); off += 8 * n_intv
            index.append((bins, linear))
        if off + 8 <= len(data):
            unmapped += struct.unpack_from("<Q", data, off)[0]
        self._index = index
        self._index_stats = (mapped, unmapped)

    @property
    def mapped(self) -> int:
        if self._index_stats is None:
            raise ValueError("mapping information not recorded in index or index not available")
        return self._index_stats[0]

    @property
    def unmapped(self) -> int:
        if self._index_stats is None:
            raise ValueError("mapping information not recorded in index or index not available")
        return self._index_stats[1]

    def gettid(self, reference: str) -> int:
        return self._tid.get(reference, -1)

    def get_reference_name(self, tid: int) -> str:
        return self.references[tid]

    def close(self):
        self._bgzf.close()

    # ---- iteration
    def _next_record(self) -> Optional[AlignedSegment]:
        szb = self._bgzf.read(4)
        if len(szb) < 4:
            return None
        size = struct.unpack("<i", szb)[0]
        data = self._bgzf.read(size)
        if len(data) < size:
            return None
        return AlignedSegment(self, data)

    def fetch(self, contig: Optional[str] = None, start=None, stop=None, **kwargs) -> Iterator[AlignedSegment]:
        """Reads overlapping [start, stop) in coordinate order; all reads when contig is None."""
        reference = kwargs.get("reference", contig)
        if "end" in kwargs and stop is None:
            stop = kwargs["end"]
        if reference is None:
            self._bgzf.seek(self._first_record)
            while True:
                r = self._next_record()
                # pysam walks an indexed file reference by reference (IteratorRowAllRefs): the unplaced unmapped
                # reads at the end of a coordinate-sorted BAM (reference id -1) are never yielded
                if r is None or r.reference_id < 0:
                    return
                yield r
        tid = self.gettid(reference)
        if tid < 0:
            raise ValueError("invalid contig `%s`" % reference)
        beg = 0 if start is None else max(0, int(start))
        end = self.lengths[tid] if stop is None else int(stop)
        if end <= beg:
            return
        if self._index is None:
            raise ValueError("fetch called on bamfile without index")
        bins, linear = self._index[tid]
        min_off = 0
        li = beg >> 14
        if linear:
            min_off = linear[li] if li < len(linear) else linear[-1]
        chunks = []
        for b in _reg2bins(beg, end):
            for cb, ce in bins.get(b, ()):
                if ce > min_off:
                    chunks.append((cb, ce))
        if not chunks:
            return
        chunks.sort()
        merged = [list(chunks[0])]
        for cb, ce in chunks[1:]:
            if cb <= merged[-1][1]:
                if ce > merged[-1][1]:
                    merged[-1][1] = ce
            else:
                merged.append([cb, ce])
        for cb, ce in merged:
            self._bgzf.seek(cb)
            while self._bgzf.tell() < ce:
                r = self._next_record()
                if r is None:
                    break
                if r.reference_id != tid or r.reference_start >= end:
                    return
                rend = r.reference_end
                if rend is None or rend <= r.reference_start:
                    rend = r.reference_start + 1
                if rend > beg:
                    yield r

    def count(self, contig=None, start=None, stop=None, read_callback="nofilter", **kwargs) -> int:
        n = 0
        for r in self.fetch(contig, start, stop, **kwargs):
            if read_callback == "all":
                if r.flag & (0x4 | 0x100 | 0x200 | 0x400):
                    continue
            n += 1
        return n


def open_alignment_file(path: str, reference_fasta: Optional[str] = None):
    """pysam.AlignmentFile when pysam is importable, else the built-in BAM reader
    (svtyper/singlesample.py:53-62 semantics: the extension decides)."""
    if not (path.endswith(".bam") or path.endswith(".cram")):
        raise ValueError("Error: %s is not a valid alignment file (*.bam or *.cram)" % path)
    try:
        import pysam  # type: ignore
        if path.endswith(".bam"):
            return pysam.AlignmentFile(path, mode="rb")
        return pysam.AlignmentFile(path, mode="rc", reference_filename=reference_fasta)
    except ImportError:
        return AlignmentFile(path, "rb")
