// svt_geometry_math.h -- the breakpoint-dependent geometry predicates of one read-fragment -> its 16-byte evidence record
// (svtyper/parsers.py:785-857, 1122-1215; the walk of svtyper/classic.py:296-396 per fragment).
//
// ONE statement of the predicates for both places that evaluate them: the device stage over fragment summaries
// (svt_geometry_kernel.h, hipcc) and the native reader when it hands over evidence records directly (svt_reads.cpp, any
// C++17 compiler: 16 bytes per fragment cross PCIe instead of a 128-byte summary).  Plain integer / binary64 arithmetic,
// no library calls: the two builds produce the same bits (tests/test_native_reads.py compares them record by record on the
// device's output).
#ifndef SVT_GEOMETRY_MATH_H
#define SVT_GEOMETRY_MATH_H

#include <stdint.h>

#include "../../include/svtyper_hip.h"

#if defined(__HIP__)     /* the translation unit is HIP source (hipcc also compiles the plain C++ files of the library) */
#define SVT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define SVT_HD inline
#endif

namespace svt {

struct ReadS { int32_t tid, start, end, iv0s, iv1s, iv0e, iv1e; uint32_t mapq, flags, extra; };
struct PieceS { int32_t tid, start, end; uint32_t mapq, flags; };
struct Record4 { uint32_t x, y, z, w; };     // svt_record as four words: ospan | mapq_a, mapq_b, rs_a, rs_b | seq_l, seq_r, clip_l, clip_r | flags

SVT_HD ReadS read_of(const svt_read_summary& s)
{
    ReadS r;
    r.tid = s.tid; r.start = s.start; r.end = s.end;
    r.iv0s = s.iv_start[0]; r.iv1s = s.iv_start[1]; r.iv0e = s.iv_end[0]; r.iv1e = s.iv_end[1];
    r.mapq = s.mapq; r.flags = s.flags; r.extra = s.reserved;
    return r;
}

SVT_HD PieceS piece_of(const svt_piece_summary& s)
{
    PieceS p;
    p.tid = s.tid; p.start = s.start; p.end = s.end; p.mapq = s.mapq; p.flags = s.flags;
    return p;
}

// parsers.py:801-816: same chromosome and get_overlap(max(0, pos - m), pos + m) >= 2 m, i.e. the
// whole 2m window lies inside one gap-free aligned interval of the read
SVT_HD bool is_ref_seq_at(const ReadS& r, int32_t tid, int32_t pos, int32_t m)
{
    if (!(r.flags & SVT_READ_PRESENT) || r.tid != tid) return false;
    if (m <= 0) return true;        // get_overlap(...) < 0 never holds
    if (pos < m) return false;      // window clipped at 0 is shorter than 2 m
    const int64_t lo = (int64_t)pos - m, hi = (int64_t)pos + m;
    return (r.iv0s <= lo && hi <= r.iv0e) || (r.iv1s <= lo && hi <= r.iv1e);
}

// one side of parsers.py:846-855
SVT_HD bool side_ok(int64_t inner, int32_t pos, int32_t ci_lo, int32_t ci_hi, bool rev, double flank)
{
    const int64_t lo = (int64_t)pos + ci_lo, hi = (int64_t)pos + ci_hi;
    if (rev) return !(inner < lo || (double)inner > (double)hi + flank);
    return !(inner > hi || (double)inner < (double)lo - flank);
}

// parsers.py:821-857
SVT_HD bool pair_straddle(const ReadS& a, const ReadS& b, bool pair_ok, int32_t tid_a, int32_t pos_a, int32_t cia_lo,
                          int32_t cia_hi, int32_t tid_b, int32_t pos_b, int32_t cib_lo, int32_t cib_hi, bool o1, bool o2,
                          int32_t m, double flank)
{
    if (!pair_ok) return false;
    if (((a.flags & SVT_READ_REVERSE) != 0) != o1 || ((b.flags & SVT_READ_REVERSE) != 0) != o2) return false;
    if (a.tid != tid_a || b.tid != tid_b) return false;
    const int64_t i1 = (int64_t)a.start + m, i2 = (int64_t)b.end - m - 1;   // get_ispan :785-789
    return side_ok(i1, pos_a, cia_lo, cia_hi, o1, flank) && side_ok(i2, pos_b, cib_lo, cib_hi, o2, flank);
}

// parsers.py:1122-1134
SVT_HD bool split_support(const PieceS& p, int32_t tid, int32_t pos, bool rev, int32_t slop)
{
    if (p.tid != tid) return false;
    const int64_t coord = rev ? p.start : p.end;
    return !(coord > (int64_t)pos + slop || coord < (int64_t)pos - slop);
}

// parsers.py:1136-1215 for one candidate; returns gated MAPQs (left | right << 8)
SVT_HD uint32_t split_weights(const PieceS& L, const PieceS& R, bool soft, const svt_breakpoint& bp, int32_t slop)
{
    if (!(L.flags & SVT_READ_PRESENT)) return 0u;
    const bool o1 = (bp.flags & SVT_BP_REV_A) != 0, o2 = (bp.flags & SVT_BP_REV_B) != 0;
    int32_t tid_lo = bp.tid_a, pos_lo = bp.pos_a, tid_hi = bp.tid_b, pos_hi = bp.pos_b;
    bool rev_lo = o1, rev_hi = o2;
    if (bp.tid_a != bp.tid_b || bp.pos_a > bp.pos_b) {   // arrange the breakends left to right (:1143-1161)
        tid_lo = bp.tid_b; pos_lo = bp.pos_b; rev_lo = o2;
        tid_hi = bp.tid_a; pos_hi = bp.pos_a; rev_hi = o1;
    }
    bool left = false, right = false;
    if (!soft || bp.svtype == SVT_SVTYPE_DEL) {           // (svtype INS never reaches the genotyper)
        left = split_support(L, tid_lo, pos_lo, rev_lo, slop);
        right = split_support(R, tid_hi, pos_hi, rev_hi, slop);
    } else if (bp.svtype == SVT_SVTYPE_DUP) {
        left = split_support(L, tid_hi, pos_hi, rev_hi, slop);
        right = split_support(R, tid_lo, pos_lo, rev_lo, slop);
    } else if (bp.svtype == SVT_SVTYPE_INV) {
        left = split_support(L, tid_lo, pos_lo, rev_lo, slop) || split_support(L, tid_hi, pos_hi, rev_hi, slop);
        right = split_support(R, tid_lo, pos_lo, rev_lo, slop) || split_support(R, tid_hi, pos_hi, rev_hi, slop);
    }
    return (left ? L.mapq : 0u) | ((right ? R.mapq : 0u) << 8);
}

// The evidence record of one fragment summary against one breakpoint.  `flank` = mean + 3 sd of the fragment's library
// (is_pair_straddle's), `m` = min_aligned, `slop` = the split slop.  The library index travels in the record.
SVT_HD Record4 geometry_record(const ReadS& ra, const ReadS& rb, const PieceS& sl, const PieceS& sr, const PieceS& cl,
                               const PieceS& cr, const svt_breakpoint& bp, double flank, int32_t m, int32_t slop)
{
    const uint32_t lib = ra.extra & 0xffffu;
    const bool pair_ok = (rb.extra & SVT_FRAG_PAIR) != 0;
    const bool cont = (rb.extra & SVT_FRAG_CONTINUATION) != 0;
    const bool o1 = (bp.flags & SVT_BP_REV_A) != 0, o2 = (bp.flags & SVT_BP_REV_B) != 0;

    // gated MAPQs of the primary reads (classic.py:306-311)
    const uint32_t rs_a = (is_ref_seq_at(ra, bp.tid_a, bp.pos_a, m) || is_ref_seq_at(ra, bp.tid_b, bp.pos_b, m)) ? ra.mapq : 0u;
    const uint32_t rs_b = (is_ref_seq_at(rb, bp.tid_a, bp.pos_a, m) || is_ref_seq_at(rb, bp.tid_b, bp.pos_b, m)) ? rb.mapq : 0u;
    // gated MAPQs of the split candidates (classic.py:317-328)
    const uint32_t wseq = split_weights(sl, sr, false, bp, slop);
    const uint32_t wclip = split_weights(cl, cr, true, bp, slop);

    // paired-end bits (classic.py:339-396), without the small-deletion gate
    uint32_t flags = (lib << SVT_REC_LIB_SHIFT) | (cont ? SVT_REC_CONTINUATION : 0u);
    uint32_t mq = 0, ospan = 0;
    if (pair_ok) {
        flags |= SVT_REC_HAS_PAIR;
        mq = ra.mapq | (rb.mapq << 8);
        const int64_t o = (int64_t)rb.end - (int64_t)ra.start;          // parsers.py:792-796,866-869
        const int64_t ao = o < 0 ? -o : o;
        ospan = (uint32_t)(ao > (int64_t)0x7fffffff ? (int64_t)0x7fffffff : ao);
        bool alt = pair_straddle(ra, rb, true, bp.tid_a, bp.pos_a, bp.ci_a[0], bp.ci_a[1], bp.tid_b, bp.pos_b, bp.ci_b[0],
                                 bp.ci_b[1], o1, o2, m, flank);
        if (!alt && bp.svtype == SVT_SVTYPE_INV)                          // reciprocal orientation (:349-357)
            alt = pair_straddle(ra, rb, true, bp.tid_a, bp.pos_a, bp.ci_a[0], bp.ci_a[1], bp.tid_b, bp.pos_b, bp.ci_b[0],
                                bp.ci_b[1], !o1, !o2, m, flank);
        if (alt) flags |= SVT_REC_ALT_STRADDLE;
        if (pair_straddle(ra, rb, true, bp.tid_a, bp.pos_a, 0, 0, bp.tid_a, bp.pos_a, 0, 0, false, true, m, flank))
            flags |= SVT_REC_REF_STRADDLE_A;                               // :387-391
        if (pair_straddle(ra, rb, true, bp.tid_b, bp.pos_b, 0, 0, bp.tid_b, bp.pos_b, 0, 0, false, true, m, flank))
            flags |= SVT_REC_REF_STRADDLE_B;                               // :392-396
    }
    Record4 out;
    out.x = ospan;
    out.y = mq | (rs_a << 16) | (rs_b << 24);
    out.z = wseq | (wclip << 16);
    out.w = flags;
    return out;
}

}  // namespace svt

#endif  // SVT_GEOMETRY_MATH_H
