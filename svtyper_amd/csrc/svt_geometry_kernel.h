// svt_geometry_kernel.h -- breakpoint-dependent geometry predicates on the device
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_GEOMETRY_KERNEL_H
#define SVT_GEOMETRY_KERNEL_H

#include "svt_device_types.h"

namespace svt {

// ------------------------------------------------------------------------------------------
// geometry kernel: one fragment summary per thread -> one canonical evidence record
// (svtyper/parsers.py:785-857, 1122-1215; the walk of svtyper/classic.py:296-396 per fragment)
// ------------------------------------------------------------------------------------------
struct GeomArgs {
    const uint4* frags;          // svt_fragment[n_frags] viewed as 8 x uint4
    const uint64_t* frag_offset; // [n_units + 1]: the unit of fragment i is found by binary search
    uint64_t n_units;
    const svt_breakpoint* bps;
    const LibDesc* libs;         // v_nondel = lib.mean + lib.sd * 3 is also is_pair_straddle's flank
    uint64_t n_frags;
    uint32_t n_libs;
    int32_t min_aligned;
    int32_t split_slop;
    uint4* records;
    uint32_t* err;
};

struct ReadS { int32_t tid, start, end, iv0s, iv1s, iv0e, iv1e; uint32_t mapq, flags, extra; };
struct PieceS { int32_t tid, start, end; uint32_t mapq, flags; };

__device__ __forceinline__ ReadS unpack_read(const uint4 a, const uint4 b)
{
    ReadS r;
    r.tid = (int32_t)a.x; r.start = (int32_t)a.y; r.end = (int32_t)a.z;
    r.iv0s = (int32_t)a.w; r.iv1s = (int32_t)b.x; r.iv0e = (int32_t)b.y; r.iv1e = (int32_t)b.z;
    r.mapq = b.w & 0xffu; r.flags = (b.w >> 8) & 0xffu; r.extra = b.w >> 16;
    return r;
}

__device__ __forceinline__ PieceS unpack_piece(const uint4 a)
{
    PieceS p;
    p.tid = (int32_t)a.x; p.start = (int32_t)a.y; p.end = (int32_t)a.z;
    p.mapq = a.w & 0xffu; p.flags = (a.w >> 8) & 0xffu;
    return p;
}

// parsers.py:801-816: same chromosome and get_overlap(max(0, pos - m), pos + m) >= 2 m, i.e. the
// whole 2m window lies inside one gap-free aligned interval of the read
__device__ __forceinline__ bool is_ref_seq_dev(const ReadS& r, int32_t tid, int32_t pos, int32_t m)
{
    if (!(r.flags & SVT_READ_PRESENT) || r.tid != tid) return false;
    if (m <= 0) return true;        // get_overlap(...) < 0 never holds
    if (pos < m) return false;      // window clipped at 0 is shorter than 2 m
    const int64_t lo = (int64_t)pos - m, hi = (int64_t)pos + m;
    return (r.iv0s <= lo && hi <= r.iv0e) || (r.iv1s <= lo && hi <= r.iv1e);
}

// one side of parsers.py:846-855
__device__ __forceinline__ bool side_ok(int64_t inner, int32_t pos, int32_t ci_lo, int32_t ci_hi, bool rev, double flank)
{
    const int64_t lo = (int64_t)pos + ci_lo, hi = (int64_t)pos + ci_hi;
    if (rev) return !(inner < lo || (double)inner > (double)hi + flank);
    return !(inner > hi || (double)inner < (double)lo - flank);
}

// parsers.py:821-857
__device__ __forceinline__ bool pair_straddle_dev(const ReadS& a, const ReadS& b, bool pair_ok, int32_t tid_a,
                                                  int32_t pos_a, int32_t cia_lo, int32_t cia_hi, int32_t tid_b,
                                                  int32_t pos_b, int32_t cib_lo, int32_t cib_hi, bool o1, bool o2,
                                                  int32_t m, double flank)
{
    if (!pair_ok) return false;
    if (((a.flags & SVT_READ_REVERSE) != 0) != o1 || ((b.flags & SVT_READ_REVERSE) != 0) != o2) return false;
    if (a.tid != tid_a || b.tid != tid_b) return false;
    const int64_t i1 = (int64_t)a.start + m, i2 = (int64_t)b.end - m - 1;   // get_ispan :785-789
    return side_ok(i1, pos_a, cia_lo, cia_hi, o1, flank) && side_ok(i2, pos_b, cib_lo, cib_hi, o2, flank);
}

// parsers.py:1122-1134
__device__ __forceinline__ bool split_support_dev(const PieceS& p, int32_t tid, int32_t pos, bool rev, int32_t slop)
{
    if (p.tid != tid) return false;
    const int64_t coord = rev ? p.start : p.end;
    return !(coord > (int64_t)pos + slop || coord < (int64_t)pos - slop);
}

// parsers.py:1136-1215 for one candidate; returns gated MAPQs (left | right << 8)
__device__ __forceinline__ uint32_t split_weights_dev(const PieceS& L, const PieceS& R, bool soft,
                                                      const svt_breakpoint& bp, int32_t slop)
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
        left = split_support_dev(L, tid_lo, pos_lo, rev_lo, slop);
        right = split_support_dev(R, tid_hi, pos_hi, rev_hi, slop);
    } else if (bp.svtype == SVT_SVTYPE_DUP) {
        left = split_support_dev(L, tid_hi, pos_hi, rev_hi, slop);
        right = split_support_dev(R, tid_lo, pos_lo, rev_lo, slop);
    } else if (bp.svtype == SVT_SVTYPE_INV) {
        left = split_support_dev(L, tid_lo, pos_lo, rev_lo, slop) || split_support_dev(L, tid_hi, pos_hi, rev_hi, slop);
        right = split_support_dev(R, tid_lo, pos_lo, rev_lo, slop) || split_support_dev(R, tid_hi, pos_hi, rev_hi, slop);
    }
    return (left ? L.mapq : 0u) | ((right ? R.mapq : 0u) << 8);
}

__global__ __launch_bounds__(kBlock) void svt_geometry_kernel(const GeomArgs g)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= g.n_frags) return;
    const uint4* __restrict__ f = g.frags + i * 8;
    const ReadS ra = unpack_read(f[0], f[1]);
    const ReadS rb = unpack_read(f[2], f[3]);
    const PieceS sl = unpack_piece(f[4]), sr = unpack_piece(f[5]);
    const PieceS cl = unpack_piece(f[6]), cr = unpack_piece(f[7]);
    uint64_t lo = 0, hi = g.n_units;               // largest u with frag_offset[u] <= i (units may be empty)
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (g.frag_offset[mid] <= i) lo = mid; else hi = mid;
    }
    const svt_breakpoint bp = g.bps[lo];
    const uint32_t lib = ra.extra & 0xffu;
    const bool pair_ok = (rb.extra & SVT_FRAG_PAIR) != 0;
    const bool cont = (rb.extra & SVT_FRAG_CONTINUATION) != 0;
    uint32_t bad = 0;
    if (lib >= g.n_libs) bad |= 4u;
    const double flank = g.libs[min(lib, g.n_libs - 1)].v_nondel;
    const int32_t m = g.min_aligned;
    const bool o1 = (bp.flags & SVT_BP_REV_A) != 0, o2 = (bp.flags & SVT_BP_REV_B) != 0;

    // gated MAPQs of the primary reads (classic.py:306-311)
    const uint32_t rs_a = (is_ref_seq_dev(ra, bp.tid_a, bp.pos_a, m) || is_ref_seq_dev(ra, bp.tid_b, bp.pos_b, m)) ? ra.mapq : 0u;
    const uint32_t rs_b = (is_ref_seq_dev(rb, bp.tid_a, bp.pos_a, m) || is_ref_seq_dev(rb, bp.tid_b, bp.pos_b, m)) ? rb.mapq : 0u;
    // gated MAPQs of the split candidates (classic.py:317-328)
    const uint32_t wseq = split_weights_dev(sl, sr, false, bp, g.split_slop);
    const uint32_t wclip = split_weights_dev(cl, cr, true, bp, g.split_slop);

    // paired-end bits (classic.py:339-396), without the small-deletion gate
    uint32_t flags = (lib << SVT_REC_LIB_SHIFT) | (cont ? SVT_REC_CONTINUATION : 0u);
    uint32_t mq = 0, ospan = 0;
    if (pair_ok) {
        flags |= SVT_REC_HAS_PAIR;
        mq = ra.mapq | (rb.mapq << 8);
        const int64_t o = (int64_t)rb.end - (int64_t)ra.start;          // parsers.py:792-796,866-869
        ospan = (uint32_t)min((int64_t)0x7fffffff, o < 0 ? -o : o);
        bool alt = pair_straddle_dev(ra, rb, true, bp.tid_a, bp.pos_a, bp.ci_a[0], bp.ci_a[1], bp.tid_b, bp.pos_b,
                                     bp.ci_b[0], bp.ci_b[1], o1, o2, m, flank);
        if (!alt && bp.svtype == SVT_SVTYPE_INV)                          // reciprocal orientation (:349-357)
            alt = pair_straddle_dev(ra, rb, true, bp.tid_a, bp.pos_a, bp.ci_a[0], bp.ci_a[1], bp.tid_b, bp.pos_b,
                                    bp.ci_b[0], bp.ci_b[1], !o1, !o2, m, flank);
        if (alt) flags |= SVT_REC_ALT_STRADDLE;
        if (pair_straddle_dev(ra, rb, true, bp.tid_a, bp.pos_a, 0, 0, bp.tid_a, bp.pos_a, 0, 0, false, true, m, flank))
            flags |= SVT_REC_REF_STRADDLE_A;                               // :387-391
        if (pair_straddle_dev(ra, rb, true, bp.tid_b, bp.pos_b, 0, 0, bp.tid_b, bp.pos_b, 0, 0, false, true, m, flank))
            flags |= SVT_REC_REF_STRADDLE_B;                               // :392-396
    }
    g.records[i] = make_uint4(ospan, mq | (rs_a << 16) | (rs_b << 24), wseq | (wclip << 16), flags);
    if (bad) atomicOr(g.err, bad);
}


}  // namespace svt

#endif  // SVT_GEOMETRY_KERNEL_H
