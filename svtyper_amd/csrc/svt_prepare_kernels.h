// svt_prepare_kernels.h -- scan / re-tile kernels that run once per batch
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_PREPARE_KERNELS_H
#define SVT_PREPARE_KERNELS_H

#include "svt_device_types.h"

namespace svt {

// ------------------------------------------------------------------------------------------
// batch preparation kernels (run once per batch, outside the genotyping pass)
// ------------------------------------------------------------------------------------------

// which sparse streams a canonical record feeds
__device__ __forceinline__ bool has_pair_entry(const uint4 w) { return (w.w & 7u) != 0u; }
__device__ __forceinline__ bool has_weight_entry(const uint4 w) { return ((w.y >> 16) | w.z) != 0u; }

// one thread per unit: validate the record contract of include/svtyper_hip.h and count the entries
// of the two sparse streams and the range of libraries the unit references
__global__ __launch_bounds__(kBlock) void svt_scan_kernel(const uint4* __restrict__ csr,
                                                          const uint64_t* __restrict__ rec_offset,
                                                          uint64_t n_units, uint32_t n_libs,
                                                          uint4* __restrict__ counts, uint32_t* err)
{
    const uint64_t u = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (u >= n_units) return;
    const uint64_t lo = rec_offset[u], hi = rec_offset[u + 1];
    uint32_t np = 0, nq = 0, bad = 0, lib_min = 0xffu, lib_max = 0u;
    for (uint64_t j = lo; j < hi; ++j) {
        const uint4 w = csr[j];
        const uint32_t f = w.w;
        lib_min = min(lib_min, SVT_REC_LIB(f));
        lib_max = max(lib_max, SVT_REC_LIB(f));
        if (!(f & SVT_REC_HAS_PAIR) &&
            (f & (SVT_REC_ALT_STRADDLE | SVT_REC_REF_STRADDLE_A | SVT_REC_REF_STRADDLE_B))) bad |= 2u;
        if (SVT_REC_LIB(f) >= n_libs) bad |= 4u;
        if (f & ~SVT_REC_FLAG_MASK) bad |= 8u;
        if ((int32_t)w.x < 0) bad |= 16u;
        np += has_pair_entry(w) ? 1u : 0u;
        nq += has_weight_entry(w) ? 1u : 0u;
    }
    if (lo == hi) lib_min = 0u;
    counts[u] = make_uint4(np, nq, lib_min, lib_max);
    if (bad) atomicOr(err, bad);
}

struct RepackArgs {
    const uint4* csr;
    const uint64_t* lane_src;   // per tile lane: first CSR record of the unit
    const uint32_t* lane_nrec;  // per tile lane: F (0 for padding lanes)
    const TileDesc* tiles;      // in storage order
    uint4* tiled;
    uint32_t n_tiles;
};

// dense layout: CSR records -> lane-interleaved rows of 16-byte records
__global__ __launch_bounds__(kBlock) void svt_repack_dense_kernel(const RepackArgs a)
{
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x % kWave;
    const uint32_t tile_idx = blockIdx.x * kWavesPerBlock + wave;
    if (tile_idx >= a.n_tiles) return;
    const TileDesc td = a.tiles[tile_idx];
    const uint64_t src = a.lane_src[td.lane_base + lane];
    const uint32_t nrec = a.lane_nrec[td.lane_base + lane];
    for (uint32_t j = 0; j < td.rows_a; ++j) {
        const uint4 w = j < nrec ? a.csr[src + j] : make_uint4(0, 0, 0, 0);
        a.tiled[td.base_a + (uint64_t)j * kWave + lane] = w;
    }
}

// split layout: CSR records -> pair-entry rows + weight-entry rows.  Entries keep the order of the
// records they come from; a record that cannot change a sum (no straddle bit / all gated MAPQs 0)
// produces no entry in that stream.
__global__ __launch_bounds__(kBlock) void svt_repack_split_kernel(const RepackArgs a)
{
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x % kWave;
    const uint32_t tile_idx = blockIdx.x * kWavesPerBlock + wave;
    if (tile_idx >= a.n_tiles) return;
    const TileDesc td = a.tiles[tile_idx];
    const uint64_t src = a.lane_src[td.lane_base + lane];
    const uint32_t nrec = a.lane_nrec[td.lane_base + lane];
    uint4* __restrict__ outp = a.tiled + td.base_a + lane;
    uint4* __restrict__ outq = a.tiled + td.base_b + lane;
    uint32_t np = 0, nq = 0;
    uint2 hold_p = make_uint2(0, 0), hold_q = make_uint2(0, 0);
    bool frag_has_q = false;  // did the current fragment already emit a weight entry?
    for (uint32_t j = 0; j < nrec; ++j) {
        const uint4 w = a.csr[src + j];
        if (!(w.w & SVT_REC_CONTINUATION)) frag_has_q = false;
        if (has_pair_entry(w)) {
            const uint2 e = make_uint2(w.x, (w.y & 0xffffu) | ((w.w & 7u) << 16) | (SVT_REC_LIB(w.w) << 24));
            if (np & 1u) outp[(uint64_t)(np >> 1) * kWave] = make_uint4(hold_p.x, hold_p.y, e.x, e.y);
            else hold_p = e;
            ++np;
        }
        if (has_weight_entry(w)) {
            // the continuation bit only survives if the entry it continues was emitted too; a
            // dropped predecessor contributed exactly +0.0 to the fragment-local sums
            const uint2 e = make_uint2((w.y >> 16) | (w.z << 16), (w.z >> 16) | (frag_has_q ? 0x10000u : 0u));
            if (nq & 1u) outq[(uint64_t)(nq >> 1) * kWave] = make_uint4(hold_q.x, hold_q.y, e.x, e.y);
            else hold_q = e;
            ++nq;
            frag_has_q = true;
        }
    }
    uint32_t rp = np >> 1, rq = nq >> 1;
    if (np & 1u) outp[(uint64_t)rp++ * kWave] = make_uint4(hold_p.x, hold_p.y, 0, 0);
    if (nq & 1u) outq[(uint64_t)rq++ * kWave] = make_uint4(hold_q.x, hold_q.y, 0, 0);
    for (; rp < td.rows_a; ++rp) outp[(uint64_t)rp * kWave] = make_uint4(0, 0, 0, 0);
    for (; rq < td.rows_b; ++rq) outq[(uint64_t)rq * kWave] = make_uint4(0, 0, 0, 0);
}


}  // namespace svt

#endif  // SVT_PREPARE_KERNELS_H
