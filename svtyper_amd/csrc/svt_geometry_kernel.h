// svt_geometry_kernel.h -- breakpoint-dependent geometry predicates on the device
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_GEOMETRY_KERNEL_H
#define SVT_GEOMETRY_KERNEL_H

#include "svt_device_types.h"
#include "svt_geometry_math.h"

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

// (the predicates themselves: svt_geometry_math.h, shared with the native reader)

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
    const uint32_t lib = ra.extra & 0xffffu;
    uint32_t bad = 0;
    if (lib >= g.n_libs) bad |= 4u;
    const double flank = g.libs[min(lib, g.n_libs - 1)].v_nondel;
    const Record4 rec = geometry_record(ra, rb, sl, sr, cl, cr, bp, flank, g.min_aligned, g.split_slop);
    g.records[i] = make_uint4(rec.x, rec.y, rec.z, rec.w);
    if (bad) atomicOr(g.err, bad);
}


}  // namespace svt

#endif  // SVT_GEOMETRY_KERNEL_H
