// svt_packed_kernel.h -- the genotype pass over PACKED evidence (include/svtyper_hip.h: svt_packed_evidence)
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
//
// Packed evidence is what a host producer emits instead of 16-byte canonical records when the bytes have to cross
// PCIe: per unit three sparse streams of small entries in 16-byte slots, unit after unit (svt_entry_formats.h has
// the entry formats; svt_pack_evidence in svtyper_hip.hip is the encoder) -- pair entries (2 bytes for the batch's
// most common MAPQ pair, 4 otherwise), reference-read entries and split / clip candidate entries (2 bytes each).
// Entries that could only add +0.0 are not stored.  ~3.2 bytes per fragment record instead of 16.
//
// The kernel is the streaming kernel's structure (svt_ring_engine.h: workgroup sort, per-wave LDS ring fed by
// LDS-DMA, one unit per lane) with the slot arithmetic of svt_unit_math.h (
// short_pair_dword / ref_read_row / candidate_row).  A unit's slots are consumed in stream order, so every tally
// sees the additions of the reference in the reference's order (classic.py:296-408, singlesample.py:246-353).
#ifndef SVT_PACKED_KERNEL_H
#define SVT_PACKED_KERNEL_H

#include "svt_ring_engine.h"

namespace svt {

struct PackedArgs {
    const uint4* slots;           // all units' slots, 16 bytes each
    const uint32_t* slot_offset;  // 3 * n_units + 1: stream k of unit u = slots [slot_offset[3 u + k], slot_offset[3 u + k + 1])
    const svt_unit* units;
    const double* pm;             // 256
    const double* l10;            // n_l10
    const Bin* bins;              // every library's n_bins + 1
    const LibDesc* libs;          // n_libs (several libraries: staged in LDS; their tables are read from bins[] through L2)
    uint32_t n_libs;
    const PairWeights* wtab;      // 32
    uint32_t n_l10;
    uint32_t total_bins;
    uint32_t common_mq;           // mapq_a | mapq_b << 8 of the one-half-word pair entries
    uint32_t lds_rings;           // byte offset of wave 0's ring (128-byte aligned)
    uint32_t l10_where;           // kL10Shared / kL10Global
    uint32_t lds_l10;
    uint32_t unit_begin;          // this launch covers units [unit_begin, unit_end)
    uint32_t unit_end;
    uint64_t n_units;
    svt_result* out;
    uint32_t result96;            // SVT_FLAG_RESULT96: `out` holds tagged 96-byte records in the workgroups' order
    uint32_t slot_begin;          // ... the first of them this launch writes
    LibDesc lib0;
    GtConsts c;
};

// MULTI: packed evidence of several libraries (library switches in the pair stream, svt_entry_formats.h)
template <bool SSO, int R, bool MULTI>
__global__ __launch_bounds__(kBlock, 3) void svt_packed_kernel(const PackedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t kUnitsPerWg = kBlock * R;
    // LDS (svt_device_types.h): pm[256] | wtab[32] | w_alt[32], w_ref[32] | the same
    // x the common pair's weight | thr[total_bins], hist[total_bins] | log10 | rings.  Entries address it by absolute offsets.
    if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();
    unsigned char* rings = smem + a.lds_rings;
    const uint32_t tid = threadIdx.x, lane = tid % kWave;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));   // wave-uniform: ring addresses stay in SGPRs
    const uint64_t wg_base = (uint64_t)a.unit_begin + (uint64_t)blockIdx.x * kUnitsPerWg;

    uint32_t beg[R], cnt[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint64_t u = wg_base + (uint32_t)j * kBlock + tid;
        beg[j] = 0u;
        cnt[j] = 0u;
        if (u < a.unit_end) {
            beg[j] = a.slot_offset[3 * u];
            cnt[j] = a.slot_offset[3 * u + 3] - beg[j];
        }
    }

    // ---- stage the tables
    for (uint32_t i = tid; i < 256; i += kBlock) reinterpret_cast<double*>(smem + kLdsPm)[i] = a.pm[i];
    if (tid < 32) {
        const PairWeights w = a.wtab[tid];
        const double pp0 = a.pm[a.common_mq & 0xffu] * a.pm[(a.common_mq >> 8) & 0xffu];   // the very product an entry would form
        reinterpret_cast<PairWeights*>(smem + kLdsWtab)[tid] = w;
        reinterpret_cast<double*>(smem + kLdsWcol)[tid] = w.w_alt;
        reinterpret_cast<double*>(smem + kLdsWcol + kWcolRef)[tid] = w.w_ref;
        reinterpret_cast<double*>(smem + kLdsWcolC)[tid] = pp0 * w.w_alt;
        reinterpret_cast<double*>(smem + kLdsWcolC + kWcolRef)[tid] = pp0 * w.w_ref;
    }
    if (MULTI) {     // the library descriptors, where the one-library form keeps its bins
        for (uint32_t i = tid; i < a.n_libs * (uint32_t)(sizeof(LibDesc) / 8); i += kBlock)
            reinterpret_cast<uint64_t*>(smem + kLdsBins)[i] = reinterpret_cast<const uint64_t*>(a.libs)[i];
    } else {
        int32_t* s_thr = reinterpret_cast<int32_t*>(smem + kLdsBins);
        uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + kLdsBins) + a.total_bins;
        for (uint32_t i = tid; i < a.total_bins; i += kBlock) {
            const Bin bn = a.bins[i];
            s_thr[i] = bn.thr;
            s_hist[i] = bn.hist;
        }
    }
    if (a.l10_where == kL10Shared) {
        double* s_l10 = reinterpret_cast<double*>(smem + a.lds_l10);
        for (uint32_t i = tid; i < a.n_l10; i += kBlock) s_l10[i] = a.l10[i];
    }
    uint4 info[R];
    wg_sort_into_tiles<R>(rings, beg, cnt, (uint32_t)min((uint64_t)kUnitsPerWg, (uint64_t)a.unit_end - wg_base), tid, lane, wave, info);

    unsigned char* ring = rings + wave * kRingBytes;
    const uint32_t ring_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const uint32_t o = lane >> 3, rr = lane & 7u;
    const uint32_t col_even = (rr ^ (o >> 1)) << 4, col_odd = col_even ^ 64u;
    const uint32_t sw16 = ((lane >> 1) & 7u) << 4, lane_block = ring_addr + lane * 128u;
    const char* slot_bytes = reinterpret_cast<const char*>(a.slots);

#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        const uint32_t first = info[r].x, n_slots = info[r].y;
        const uint32_t unit = info[r].z == kPadUnit ? kPadUnit : (uint32_t)wg_base + info[r].z;
        svt_unit U{};
        uint32_t end_pairs = 0, end_refs = 0;    // stream boundaries, in slots from the start of the unit's first block
        const uint32_t head = first & 7u, last = head + n_slots;
        if (unit != kPadUnit) {
            U = a.units[unit];
            end_pairs = head + (a.slot_offset[3 * (uint64_t)unit + 1] - first);
            end_refs = head + (a.slot_offset[3 * (uint64_t)unit + 2] - first);
        }
        const uint32_t nblk = n_slots ? (last + 7u) >> 3 : 0u;
        uint32_t max_blk, min_blk;
        tile_block_range(nblk, max_blk, min_blk);
        uint32_t src_first[8], src_end[8], src_base[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            src_first[i] = (uint32_t)__shfl((int)first, 8 * i + (int)o, kWave);
            src_end[i] = src_first[i] + (uint32_t)__shfl((int)n_slots, 8 * i + (int)o, kWave);
            src_base[i] = (src_first[i] & ~7u) + (((i & 1) ? col_odd : col_even) >> 4);
        }

        LaneCtx c{};
        c.is_del = U.svtype == SVT_SVTYPE_DEL;
        c.del16 = c.is_del ? 16u : 0u;
        c.wt0 = kLdsWcol + c.del16 * 8u;
        c.wt1 = c.wt0 + 8u * 8u;
        c.nb4 = a.lib0.n_bins * 4u;
        c.off2_4 = c.is_del ? min((uint32_t)U.var_length, a.lib0.n_bins) * 4u : 0x80000000u;
        c.hist_at = kLdsBins + a.total_bins * 4u;
        c.common_mq = a.common_mq;
        c.var_length = U.var_length;
        c.tab8 = a.lib0.tab_off * (uint32_t)sizeof(Bin);      // several libraries: every unit's stream starts in library 0's context
        c.libs_at = kLdsBins;
        c.lib_last = a.n_libs - 1u;
        Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

        if (max_blk) {
            fetch_block<SVT_STREAM_AUX, true>(0, src_base, src_first, src_end, slot_bytes, ring);
            u32x4 w[8];
#pragma unroll 1
            for (uint32_t k = 0; k < max_blk; ++k) {
                read_block(lane_block, sw16, w);
                if (k + 1 < max_blk) fetch_block<SVT_STREAM_AUX, false>(k + 1, src_base, src_first, src_end, slot_bytes, ring);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t idx = k * kBlockRecords + (uint32_t)j;
                    const bool mine = idx >= head && idx < last;   // other slots were not fetched
                    const bool is_pair = mine && idx < end_pairs, is_ref = mine && idx >= end_pairs && idx < end_refs,
                               is_cand = mine && idx >= end_refs;
                    // a zero slot is a no-op in every stream: its entries carry MAPQ 0 / no straddle bit and add +0.0
                    if (__any(is_pair)) {
                        const uint32_t x = is_pair ? w[j].x : 0u, y = is_pair ? w[j].y : 0u, z = is_pair ? w[j].z : 0u,
                                       v = is_pair ? w[j].w : 0u;
                        short_pair_dword<MULTI>(x, c, acc, a.bins);
                        short_pair_dword<MULTI>(y, c, acc, a.bins);
                        short_pair_dword<MULTI>(z, c, acc, a.bins);
                        short_pair_dword<MULTI>(v, c, acc, a.bins);
                    }
                    if (__any(is_ref))
                        ref_read_row<SSO>(is_ref ? make_uint4(w[j].x, w[j].y, w[j].z, w[j].w) : make_uint4(0, 0, 0, 0), acc);
                    if (__any(is_cand))
                        candidate_row<SSO>(is_cand ? make_uint4(w[j].x, w[j].y, w[j].z, w[j].w) : make_uint4(0, 0, 0, 0), acc);
                }
            }
        }
        if (SSO) {  // flush the last fragment (singlesample.py:370-372)
            acc.ref_seq += acc.l_ref_seq;
            acc.alt_seq += acc.l_alt_seq;
            acc.alt_clip += acc.l_alt_clip;
        }
        uint4 piece[8];
        unit_epilogue(acc, (uint32_t)U.svtype, (uint32_t)U.flags, a.c, reinterpret_cast<const double*>(smem + a.lds_l10), a.l10,
                      a.l10_where == kL10Shared ? a.n_l10 : 0u, piece);
        // svt_result96: {GQ, GT, unit} behind QR / QA; the tile's records leave in the tile's own order (svt_stream_kernel.h)
        uint32_t tile_slot = 0xFFFFFFFFu;
        if (a.result96) {
            piece[5] = make_uint4(piece[5].x, piece[7].y, unit, 0u);
            tile_slot = a.slot_begin + blockIdx.x * (uint32_t)(kBlock * R) + ((uint32_t)r * kWavesPerBlock + ((r & 1) ? (uint32_t)kWavesPerBlock - 1u - wave : wave)) * kWave;
        }
        store_result_records_through_ring(ring, piece, unit, lane, reinterpret_cast<unsigned char*>(a.out), a.result96 ? 6u : 8u, tile_slot);
    }
}

}  // namespace svt

#endif  // SVT_PACKED_KERNEL_H
