// svt_stream_kernel.h -- the genotype pass over the canonical CSR records as they are (no re-tiling)
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
//
// One kernel launch takes the section-8(d) canonical input -- rec_offset[], svt_unit[], svt_record[] in the
// order the caller packed them -- to the 128-byte result records: tally -> zeroing rules -> QR/QA ->
// bayes_gt -> GT/GQ/SQ (svtyper/classic.py:296-513, singlesample.py:246-473).  Every record is read from
// HBM exactly once, nothing is re-encoded, sorted or tiled on the host or in a separate kernel.
//
// The five tallies are sequential binary64 sums in record order, so a unit still belongs to ONE lane.
// What makes that coalesced here is a per-wave LDS ring filled by LDS-DMA (global_load_lds_dwordx4):
//
//   * a workgroup owns 256 * R consecutive units; it counting-sorts them by their number of 128-byte
//     record blocks (LDS atomics + one scan) so that the 64 lanes of a wave run units of similar
//     length, and hands the sorted 64-unit tiles to its four waves in snake order (wave w: tiles w,
//     7 - w, 8 + w, ...), which balances the waves of the workgroup;
//   * per step a wave fetches, for each of its 64 units, the next 128-byte block (8 records) of that
//     unit: eight LDS-DMA instructions, each serving eight units with eight lanes per unit, so every
//     instruction moves eight whole cache lines and nothing passes through VGPRs;
//   * the block of unit u lands at ring + u * 128 with its eight 16-byte slots XOR-swizzled by
//     (u >> 1) & 7 -- the lane that owns unit u then reads its records with ds_read_b128 and the 16
//     lanes the LDS serves per cycle hit 16 different bank quads (conflict-free, MI355X_MICROARCH LDS table);
//   * two stages per wave (2 x 8 KB): block k + 2 is in flight while block k is consumed;
//   * records of a block that lie outside the unit (the neighbours' records in its first and last
//     block) are consumed with their weight bytes zeroed: prob_mapq(0) == +0.0 exactly, and x + 0.0 == x
//     for these non-negative sums (the argument of include/svtyper_hip.h for gated-off reads);
//   * the result records leave through the same ring: each lane stores its eight pieces to LDS, the wave
//     reads them back unit-major and every group of eight lanes writes one full 128-byte line.
//
// The record contract of include/svtyper_hip.h is checked on the fly (svt_scan_kernel's job for the
// tiled layouts): violations are OR-ed into *err, which the host reads after the pass.
#ifndef SVT_STREAM_KERNEL_H
#define SVT_STREAM_KERNEL_H

#include "svt_genotype_kernel.h"

namespace svt {

constexpr uint32_t kBlockRecords = 8;                         // records per 128-byte block
constexpr uint32_t kStageBytes = kWave * 128;                 // one block per lane
constexpr uint32_t kRingStages = 2;
constexpr uint32_t kRingBytes = kStageBytes * kRingStages;    // per wave
constexpr uint32_t kLdsStreamBins = kLdsWtab + 32 * 16;       // Bin[lds_bins], then LibDesc[lds_libs], then the rings
constexpr uint32_t kMaxSortKey = 255;                         // units with more blocks share the last sort bucket

// error bits (shared with svt_scan_kernel)
constexpr uint32_t kErrStraddleNoPair = 2u, kErrLibIndex = 4u, kErrReservedBits = 8u, kErrNegativeSpan = 16u;

struct StreamArgs {
    const uint4* records;        // canonical records; the allocation ends on a 128-byte block boundary, tail zeroed
    const uint64_t* rec_offset;  // n_units + 1
    const svt_unit* units;
    const double* pm;            // 256
    const double* l10;           // n_l10, allocation padded to whole KiB
    const LibDesc* libs;
    const Bin* bins;
    const PairWeights* wtab;     // 32
    uint32_t n_l10;
    uint32_t n_libs;
    uint32_t total_bins;
    uint32_t last_blk;           // index of the last 128-byte block of the records
    uint32_t lds_bins;           // bins staged in LDS (kSingleLds: the whole table)
    uint32_t lds_libs;           // library descriptors staged in LDS
    uint32_t lds_rings;          // byte offset of wave 0's ring (128-byte aligned)
    uint32_t l10_in_ring;        // the log10 table fits the ring (n_l10 * 8 <= kRingBytes)
    uint64_t n_units;
    svt_result* out;
    uint32_t* err;
    LibDesc lib0;
    GtConsts c;
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;

// eight consecutive records of this lane's block from stage STAGE of the ring, once the LDS-DMA group that
// filled it has landed: all but the youngest PENDING vector-memory operations must have completed.  The
// compiler cannot see that these reads depend on the LDS-DMA writes, hence the explicit counters.
template <int STAGE, int PENDING>
__device__ __forceinline__ void read_block(const uint32_t (&addr)[8], u32x4 (&w)[8])
{
    asm volatile("s_waitcnt vmcnt(%[pend])\n\t"
                 "ds_read_b128 %0, %8 offset:%[off]\n\t"
                 "ds_read_b128 %1, %9 offset:%[off]\n\t"
                 "ds_read_b128 %2, %10 offset:%[off]\n\t"
                 "ds_read_b128 %3, %11 offset:%[off]\n\t"
                 "ds_read_b128 %4, %12 offset:%[off]\n\t"
                 "ds_read_b128 %5, %13 offset:%[off]\n\t"
                 "ds_read_b128 %6, %14 offset:%[off]\n\t"
                 "ds_read_b128 %7, %15 offset:%[off]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                 : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),
                   [pend] "n"(PENDING), [off] "n"(STAGE * (int)kStageBytes)
                 : "memory");
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, const uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, d, kWave);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

struct RecordCheck {      // accumulated over every record a lane sees (neighbours' included: they are part of the batch)
    uint32_t undefined = 0, span = 0, lone_straddle = 0, max_lib = 0;
    __device__ __forceinline__ void see(const u32x4 w)
    {
        undefined |= w.w & ~SVT_REC_FLAG_MASK;
        span |= w.x;                                                    // sign bit: a negative ospan_len
        lone_straddle |= (w.w & 7u) & (((w.w >> 4) & 1u) - 1u);          // straddle bits without HAS_PAIR
        max_lib = max(max_lib, (w.w >> SVT_REC_LIB_SHIFT) & 0xffu);
    }
    __device__ __forceinline__ uint32_t bits(const uint32_t n_libs) const
    {
        return (lone_straddle ? kErrStraddleNoPair : 0u) | (max_lib >= n_libs ? kErrLibIndex : 0u) |
               (undefined ? kErrReservedBits : 0u) | ((int32_t)span < 0 ? kErrNegativeSpan : 0u);
    }
};

template <bool SSO, int MODE, int R>
__global__ __launch_bounds__(kBlock) void svt_stream_kernel(const StreamArgs a)
{
    static_assert(MODE == kSingleLds || MODE == kGeneral, "library windows are not used by the streaming kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // (lds_rings is 128-byte aligned)
    constexpr uint32_t kUnitsPerWg = kBlock * R;
    constexpr uint32_t kTilesPerWg = kWavesPerBlock * R;
    double* s_pm = reinterpret_cast<double*>(smem + kLdsPm);
    PairWeights* s_wtab = reinterpret_cast<PairWeights*>(smem + kLdsWtab);
    Bin* s_bins = reinterpret_cast<Bin*>(smem + kLdsStreamBins);
    LibDesc* s_lib = reinterpret_cast<LibDesc*>(s_bins + a.lds_bins);
    unsigned char* rings = smem + a.lds_rings;
    // sort scratch: lives in the rings until the streaming starts
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(rings);      // kMaxSortKey + 1 buckets
    uint32_t* s_start = s_hist + (kMaxSortKey + 1);
    uint32_t* s_wsum = s_start + (kMaxSortKey + 1);              // kWavesPerBlock
    uint4* s_info = reinterpret_cast<uint4*>(s_wsum + 8);        // per sorted position: {first record, records, local unit, -}

    const uint32_t tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
    const uint64_t wg_base = (uint64_t)blockIdx.x * kUnitsPerWg;

    // ---- this thread's R units: record range and sort key (the loads overlap the table staging below)
    uint32_t beg[R], cnt[R], key[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint64_t u = wg_base + (uint32_t)j * kBlock + tid;
        beg[j] = 0u;
        cnt[j] = 0u;
        if (u < a.n_units) {
            const uint64_t lo = a.rec_offset[u], hi = a.rec_offset[u + 1];
            beg[j] = (uint32_t)lo;
            cnt[j] = (uint32_t)(hi - lo);
        }
        const uint32_t nblk = cnt[j] ? ((beg[j] & 7u) + cnt[j] + 7u) >> 3 : 0u;
        key[j] = min(nblk, kMaxSortKey);
    }

    // ---- stage the tables in LDS
    for (uint32_t i = tid; i < 256; i += kBlock) s_pm[i] = a.pm[i];
    if (tid < 32) s_wtab[tid] = a.wtab[tid];
    for (uint32_t i = tid; i < a.lds_libs * (uint32_t)(sizeof(LibDesc) / 8); i += kBlock)
        reinterpret_cast<uint64_t*>(s_lib)[i] = reinterpret_cast<const uint64_t*>(a.libs)[i];
    for (uint32_t i = tid; i < a.lds_bins; i += kBlock)
        reinterpret_cast<uint64_t*>(s_bins)[i] = reinterpret_cast<const uint64_t*>(a.bins)[i];
    for (uint32_t i = tid; i <= kMaxSortKey; i += kBlock) s_hist[i] = 0u;
    __syncthreads();

    // ---- counting sort of the workgroup's units by block count, longest first
    uint32_t rank[R];
#pragma unroll
    for (int j = 0; j < R; ++j) rank[j] = atomicAdd(&s_hist[key[j]], 1u);
    __syncthreads();
    {
        // thread t owns bucket kMaxSortKey - t (kBlock == kMaxSortKey + 1): an exclusive scan over t is the
        // first sorted position of every bucket in descending key order
        static_assert(kBlock == (int)kMaxSortKey + 1, "one sort bucket per thread");
        const uint32_t h = s_hist[kMaxSortKey - tid];
        const uint32_t incl = wave_inclusive_scan(h, lane);
        if (lane == kWave - 1) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) before += (uint32_t)w < wave ? s_wsum[w] : 0u;
        s_start[kMaxSortKey - tid] = before + incl - h;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint64_t u = wg_base + (uint32_t)j * kBlock + tid;
        s_info[s_start[key[j]] + rank[j]] = make_uint4(beg[j], cnt[j], u < a.n_units ? (uint32_t)j * kBlock + tid : kPadUnit, 0u);
    }
    __syncthreads();
    // the r-th tile of this wave in snake order
    uint4 info[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t tile = (uint32_t)r * kWavesPerBlock + ((r & 1) ? (uint32_t)kWavesPerBlock - 1u - wave : wave);
        info[r] = s_info[tile * kWave + lane];
    }
    static_assert(kTilesPerWg * kWave == kUnitsPerWg, "tiles cover the workgroup's units");
    __syncthreads();   // the rings are free from here on

    Tables t;
    t.pm = s_pm;
    t.wtab = s_wtab;
    t.libs = s_lib;
    t.bins = MODE == kSingleLds ? s_bins : a.bins;

    unsigned char* ring = rings + wave * kRingBytes;
    const uint32_t ring_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    // LDS-DMA: lane (o, rr) of instruction i fetches 16 bytes of the block of unit 8 i + o; they land at
    // ring + (8 i + o) * 128 + rr * 16.  The slot rr of unit u holds logical record rr ^ swz(u), swz(u) = (u >> 1) & 7
    // = (o >> 1) | (i & 1) << 2.
    const uint32_t o = lane >> 3, rr = lane & 7u;
    const uint32_t col_even = (rr ^ (o >> 1)) << 4, col_odd = col_even ^ 64u;
    // consumer: logical record j of this lane's block
    const uint32_t sw = (lane >> 1) & 7u;
    uint32_t rd_addr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rd_addr[j] = ring_addr + lane * 128u + (((uint32_t)j ^ sw) << 4);
    const char* rec_bytes = reinterpret_cast<const char*>(a.records);

    RecordCheck check;

#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        const uint32_t first_rec = info[r].x, n_rec = info[r].y;
        const uint32_t unit = info[r].z == kPadUnit ? kPadUnit : (uint32_t)wg_base + info[r].z;   // n_units < 2^32
        svt_unit U{};
        if (unit != kPadUnit) U = a.units[unit];
        const uint32_t head = first_rec & 7u, last = head + n_rec;
        const uint32_t blk0 = first_rec >> 3;
        const uint32_t nblk = n_rec ? (last + 7u) >> 3 : 0u;
        // sorted longest first: the tile's first lane has the most blocks -- unless it sits in the last sort
        // bucket, which holds every longer unit in arrival order
        uint32_t max_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)nblk);
        if (max_blk >= kMaxSortKey) {
            uint32_t m = nblk;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, kWave));
            max_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
        }
        uint32_t src_blk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) src_blk[i] = (uint32_t)__shfl((int)blk0, 8 * i + (int)o, kWave);

        auto fetch = [&](const uint32_t k, const uint32_t stage) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // lanes past the end of their unit re-read blocks the neighbouring units need anyway
                const uint32_t blk = min(src_blk[i] + k, a.last_blk);
                const char* src = rec_bytes + ((uint64_t)blk << 7) + ((i & 1) ? col_odd : col_even);
                __builtin_amdgcn_global_load_lds(src, (lds_void_ptr)(ring + stage * kStageBytes + (uint32_t)i * 1024u), 16, 0, 0);
            }
        };

        LaneCtx c{};
        c.is_del = U.svtype == SVT_SVTYPE_DEL;
        c.del16 = c.is_del ? 16u : 0u;
        c.var_length = U.var_length;
        c.pos_delta_d = (double)U.pos_delta;
        {
            const bool small_del = c.is_del && (c.pos_delta_d < a.lib0.sd2);  // classic.py:339,383
            c.fmask = small_del ? 0u : 7u;
            c.kmin = (uint32_t)a.lib0.key_min;
            c.nb = a.lib0.n_bins;
            c.sub2 = c.is_del ? (uint32_t)U.var_length + (uint32_t)a.lib0.key_min : 0x80000000u;
        }
        Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

        auto consume = [&](const u32x4 (&w)[8], const uint32_t k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                check.see(w[j]);
                const uint32_t idx = k * kBlockRecords + (uint32_t)j;
                const bool mine = idx >= head && idx < last;
                const uint32_t wy = mine ? w[j].y : 0u, wz = mine ? w[j].z : 0u;   // MAPQ 0 everywhere: adds +0.0
                weight_evidence<SSO>(wy >> 16 | (wz << 16), wz >> 16, (w[j].w & SVT_REC_CONTINUATION) != 0, t, acc);
                pair_evidence<MODE>(w[j].x, wy & 0xffffu, w[j].w & 7u, min(SVT_REC_LIB(w[j].w), a.n_libs - 1u), t, c, acc);   // (a bad index is reported through *err)
            }
        };

        if (max_blk) {
            fetch(0, 0);
            if (max_blk > 1) fetch(1, 1);
            u32x4 w[8];
            for (uint32_t k = 0; k < max_blk; k += 2) {
                if (k + 1 < max_blk) read_block<0, 8>(rd_addr, w);
                else read_block<0, 0>(rd_addr, w);
                if (k + 2 < max_blk) fetch(k + 2, 0);
                consume(w, k);
                if (k + 1 >= max_blk) break;
                if (k + 2 < max_blk) read_block<1, 8>(rd_addr, w);
                else read_block<1, 0>(rd_addr, w);
                if (k + 3 < max_blk) fetch(k + 3, 1);
                consume(w, k + 1);
            }
        }
        if (SSO) {  // flush the last fragment (singlesample.py:370-372)
            acc.ref_seq += acc.l_ref_seq;
            acc.alt_seq += acc.l_alt_seq;
            acc.alt_clip += acc.l_alt_clip;
        }

        // ---- epilogue: the log10 table of log_choose goes through the (now idle) ring when it fits
        double* ring_l10 = reinterpret_cast<double*>(ring);
        if (a.l10_in_ring) {
            const char* l10_bytes = reinterpret_cast<const char*>(a.l10);
            for (uint32_t off = 0; off < a.n_l10 * 8u; off += 1024u)
                __builtin_amdgcn_global_load_lds(l10_bytes + off + lane * 16u, (lds_void_ptr)(ring + off), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        uint4 piece[8];
        unit_epilogue(acc, (uint32_t)U.svtype, (uint32_t)U.flags, a.c, ring_l10, a.l10, a.l10_in_ring != 0u, piece);

        // ---- result records: lane-major into the ring, unit-major out of it, one full line per eight lanes
        uint4* st = reinterpret_cast<uint4*>(ring + lane * 128u);
#pragma unroll
        for (int p = 0; p < 8; ++p) st[(uint32_t)p ^ sw] = piece[p];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t dst_unit = (uint32_t)__shfl((int)unit, 8 * i + (int)o, kWave);
            const uint4 v = *reinterpret_cast<const uint4*>(ring + (uint32_t)i * 1024u + o * 128u + ((i & 1) ? col_odd : col_even));
            if (dst_unit != kPadUnit) reinterpret_cast<uint4*>(a.out + dst_unit)[rr] = v;
        }
    }
    const uint32_t bad = check.bits(a.n_libs);
    if (bad) atomicOr(a.err, bad);
}

}  // namespace svt

#endif  // SVT_STREAM_KERNEL_H
