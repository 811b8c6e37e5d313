"""Test helper: write a small coordinate-sorted BAM + BAI from Python dict records (BGZF blocks cut at a
fixed size regardless of record boundaries, so records span blocks).  Only what the tests need: no
sequence / quality content (SEQ is written as '=' nibbles, QUAL as 0xff)."""
import struct
import zlib

CIGAR_OPS = "MIDNSHP=X"


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def parse_cigar(text):
    out, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            out.append((CIGAR_OPS.index(ch), int(num)))
            num = ""
    return out


def ref_len(cigar):
    return sum(n for op, n in cigar if op in (0, 2, 3, 7, 8))


def query_len(cigar):
    return sum(n for op, n in cigar if op in (0, 1, 4, 7, 8))


def encode_record(r):
    """r: dict(name, flag, tid, pos, mapq, cigar (text), mtid, mpos, tlen, tags: list of (key, type, value))"""
    cigar = parse_cigar(r["cigar"]) if r["cigar"] != "*" else []
    l_seq = query_len(cigar)
    end = r["pos"] + max(1, ref_len(cigar))
    name = r["name"].encode() + b"\0"
    body = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(name), r["mapq"], reg2bin(r["pos"], end), len(cigar),
                       r["flag"], l_seq, r["mtid"], r["mpos"], r["tlen"])
    body += name
    body += b"".join(struct.pack("<I", (n << 4) | op) for op, n in cigar)
    if "seq4" in r:     # optional real content: packed 4-bit bases + qualities (realistic inflate cost)
        assert len(r["seq4"]) == (l_seq + 1) // 2 and len(r["qual"]) == l_seq
        body += r["seq4"] + r["qual"]
    else:
        body += b"\x00" * ((l_seq + 1) // 2) + b"\xff" * l_seq
    for key, typ, val in r.get("tags", ()):
        if typ == "raw":        # bytes as they are (a malformed tag for the readers' error paths)
            body += val
            continue
        body += key.encode() + typ.encode()
        if typ == "Z":
            body += val.encode() + b"\0"
        elif typ == "C":
            body += struct.pack("<B", val)
        elif typ == "i":
            body += struct.pack("<i", val)
        elif typ == "B":   # (subtype, values)
            sub, vals = val
            code = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
            body += sub.encode() + struct.pack("<I", len(vals)) + b"".join(struct.pack("<" + code, v) for v in vals)
        else:
            raise ValueError(typ)
    return struct.pack("<i", len(body)) + body, end


def bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
            + cdata + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def write_bam(path, header_text, references, records, block_bytes=3000):
    """references: list of (name, length); records: dicts sorted by (tid, pos).  Writes path and path + '.bai'."""
    text = header_text.encode()
    stream = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(references))
    for name, length in references:
        nm = name.encode() + b"\0"
        stream += struct.pack("<i", len(nm)) + nm + struct.pack("<i", length)
    spans = []   # (stream start, stream end, tid, pos, end)
    parts = [stream]
    at = len(stream)
    for r in records:
        enc, end = encode_record(r)
        spans.append((at, at + len(enc), r["tid"], r["pos"], end))
        parts.append(enc)
        at += len(enc)
    stream = b"".join(parts)
    block_coff, coff, out = [], 0, []
    for i in range(0, len(stream), block_bytes):
        blk = bgzf_block(stream[i:i + block_bytes])
        block_coff.append(coff)
        out.append(blk)
        coff += len(blk)
    block_coff.append(coff)   # the EOF block
    out.append(BGZF_EOF)
    with open(path, "wb") as f:
        f.write(b"".join(out))

    def voff(p):
        return (block_coff[p // block_bytes] << 16) | (p % block_bytes)

    n_ref = len(references)
    bins = [dict() for _ in range(n_ref)]
    linear = [dict() for _ in range(n_ref)]
    for s, e, tid, pos, end in spans:
        if tid < 0:
            continue
        b = reg2bin(pos, end)
        chunks = bins[tid].setdefault(b, [])
        v0, v1 = voff(s), voff(e)
        if chunks and chunks[-1][1] == v0:
            chunks[-1][1] = v1
        else:
            chunks.append([v0, v1])
        for w in range(pos >> 14, ((end - 1) >> 14) + 1):
            linear[tid][w] = min(linear[tid].get(w, v0), v0)
    bai = b"BAI\x01" + struct.pack("<i", n_ref)
    for tid in range(n_ref):
        bai += struct.pack("<i", len(bins[tid]))
        for b, chunks in sorted(bins[tid].items()):
            bai += struct.pack("<Ii", b, len(chunks)) + b"".join(struct.pack("<QQ", c0, c1) for c0, c1 in chunks)
        n_intv = (max(linear[tid]) + 1) if linear[tid] else 0
        bai += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):   # windows without a record inherit the previous offset (samtools convention)
            last = linear[tid].get(w, last)
            bai += struct.pack("<Q", last)
    with open(path + ".bai", "wb") as f:
        f.write(bai)
