// svt_ring_engine.h -- per-wave LDS ring fed by LDS-DMA: the machinery that lets ONE lane own a unit whose
// 16-byte items (evidence records, or the slots of packed evidence) lie contiguously in HBM, in caller order.
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
//
//   * a workgroup owns 256 * R consecutive units; wg_sort_into_tiles counting-sorts them by their number of
//     128-byte blocks (LDS atomics + one scan) so that the 64 lanes of a wave run units of similar length, and
//     hands the sorted 64-unit tiles to the four waves in snake order;
//   * fetch_block: per step a wave fetches, for each of its 64 units, the next 128-byte block (8 items) of that
//     unit with eight global_load_lds_dwordx4 -- each instruction serves eight units with eight lanes per unit, so
//     it moves eight whole cache lines and nothing passes through VGPRs.  A lane only asks for an item of its
//     unit: the neighbours' items in a unit's first and last line, and every block past the end of a shorter
//     unit, are not requested at all;
//   * the block of unit u lands at ring + u * 128 with its eight 16-byte slots XOR-swizzled by (u >> 1) & 7:
//     read_block then takes the lane's eight items with ds_read_b128 and the 16 lanes the LDS serves per cycle
//     hit 16 different bank quads (conflict-free, MI355X_MICROARCH LDS table);
//   * one 8 KB stage per wave: a block leaves the stage for VGPRs in one burst, the fetch of block k + 1 is
//     issued right behind it and lands while block k is being consumed.
#ifndef SVT_RING_ENGINE_H
#define SVT_RING_ENGINE_H

#include "svt_unit_math.h"

namespace svt {

constexpr uint32_t kBlockRecords = 8;                         // records per 128-byte block
constexpr uint32_t kStageBytes = kWave * 128;                 // one block per lane
constexpr uint32_t kRingBytes = kStageBytes;                  // per wave: one stage
#ifndef SVT_READ_ADDR_RECOMPUTE
#define SVT_READ_ADDR_RECOMPUTE 0   // (in-process A/B: the recomputation costs 0.8 % of the one-library pass and 4 % of the window pass)
#endif
#ifndef SVT_STREAM_DEPTH
#define SVT_STREAM_DEPTH 1   // stages per wave in svt_stream_kernel (2: block k + 2 is in flight while block k is summed)
#endif
constexpr uint32_t kStreamDepth = SVT_STREAM_DEPTH;
constexpr uint32_t kStreamRingBytes = kStageBytes * kStreamDepth;
constexpr uint32_t kMaxSortKey = 255;                         // units with more blocks share the last sort bucket
// where the epilogue finds the log10 table of log_choose
enum L10Place : uint32_t {
    kL10Shared = 0,   // staged once per workgroup beside the other tables (fits with 3 workgroups per CU)
    kL10Ring = 1,     // copied into the wave's idle ring before each epilogue
    kL10Global = 2    // read through L2 (units with thousands of records)
};


typedef __attribute__((address_space(3))) void* lds_void_ptr;

// The eight records of this lane's block, once the LDS-DMA group that filled the ring has landed (it is the
// only vector-memory work the wave has in flight).  The compiler cannot see that these reads depend on the
// LDS-DMA writes, hence the explicit counters; when this returns the ring is free for the next block.
// STAGE_OFF: byte offset of the stage inside the wave's ring (immediate of the ds_reads); PENDING: vector-memory
// instructions that may still be outstanding -- the LDS-DMA group of the NEXT block, when the ring has two stages and
// that group is known to be exactly eight instructions.
template <uint32_t STAGE_OFF = 0, int PENDING = 0>
__device__ __forceinline__ void read_block(const uint32_t lane_block, const uint32_t sw16, u32x4 (&w)[8])
{
    // logical record j of the lane's block sits in slot j ^ swz: lane_block + ((j << 4) ^ sw16)
    uint32_t addr[8];
#if SVT_READ_ADDR_RECOMPUTE
    // (the eight addresses are loop invariants the compiler would keep in eight VGPRs across the block loop; formed anew
    // per block -- one v_xad_u32 each -- they live for the duration of the burst only)
    uint32_t sw = sw16;
    asm volatile("" : "+v"(sw));
#else
    const uint32_t sw = sw16;
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) addr[j] = lane_block + (((uint32_t)j << 4) ^ sw);
    asm volatile("s_waitcnt vmcnt(%16)\n\t"
                 "ds_read_b128 %0, %8 offset:%17\n\t"
                 "ds_read_b128 %1, %9 offset:%17\n\t"
                 "ds_read_b128 %2, %10 offset:%17\n\t"
                 "ds_read_b128 %3, %11 offset:%17\n\t"
                 "ds_read_b128 %4, %12 offset:%17\n\t"
                 "ds_read_b128 %5, %13 offset:%17\n\t"
                 "ds_read_b128 %6, %14 offset:%17\n\t"
                 "ds_read_b128 %7, %15 offset:%17\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                 : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),
                   "n"(PENDING), "n"(STAGE_OFF)
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

// Block k of every unit of the tile -> ring.  Lane (o, rr) of instruction i serves unit 8 i + o and only asks for
// a record of that unit: the neighbours' records in a unit's first and last line, and every block past the end
// of a shorter unit, are not requested at all (the consumer never looks at those slots).
//   src_base[i] = the item this lane would fetch from block 0 of its unit: (first item & ~7) + its column
//   FIRST: k == 0 -- only a unit's first block can hold items in front of the unit, so only that one needs the
//   lower bound (and src_first[] need not stay in registers over the block loop)
template <int AUX, bool FIRST>
__device__ __forceinline__ void fetch_block(const uint32_t k, const uint32_t (&src_base)[8], const uint32_t (&src_first)[8],
                                            const uint32_t (&src_end)[8], const char* __restrict__ rec_bytes, unsigned char* ring)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t rec = src_base[i] + (FIRST ? 0u : k * kBlockRecords);
        if ((!FIRST || rec >= src_first[i]) && rec < src_end[i])
            __builtin_amdgcn_global_load_lds(rec_bytes + ((uint64_t)rec << 4), (lds_void_ptr)(ring + (uint32_t)i * 1024u), 16, 0, AUX);
    }
}

// Block k of every unit of the tile when k is an interior block of ALL of them (1 <= k, k + 1 < the tile's shortest
// unit): every piece of every line belongs to its unit, so there is nothing to test -- eight address computations
// and eight LDS-DMA instructions with the full wave.  Most steps of a length-sorted tile are of this kind.
template <int AUX>
__device__ __forceinline__ void fetch_block_interior(const uint32_t k, const uint32_t (&src_base)[8], const char* __restrict__ rec_bytes,
                                                     unsigned char* ring)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t rec = src_base[i] + k * kBlockRecords;
        __builtin_amdgcn_global_load_lds(rec_bytes + ((uint64_t)rec << 4), (lds_void_ptr)(ring + (uint32_t)i * 1024u), 16, 0, AUX);
    }
}

// The same for a step in which some unit of the tile reaches its last block: only the lanes whose block IS their unit's
// last one -- the line the unit shares with its successor in the CSR, which another lane requests at another time --
// take the cache policy EDGE_AUX (so that the second request can be served by L2); every other line of the step is
// read once and leaves with AUX.  (With one policy for the whole step a third of all lines were allocated in L2,
// which pushed the shared lines out again before their second request came.)
template <int AUX, int EDGE_AUX>
__device__ __forceinline__ void fetch_block_tail_exact(const uint32_t k, const uint32_t (&src_base)[8], const uint32_t (&src_end)[8],
                                                       const char* __restrict__ rec_bytes, unsigned char* ring)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t rec = src_base[i] + k * kBlockRecords;
        if (rec < src_end[i]) {
            if ((rec | 7u) + 1u >= src_end[i])
                __builtin_amdgcn_global_load_lds(rec_bytes + ((uint64_t)rec << 4), (lds_void_ptr)(ring + (uint32_t)i * 1024u), 16, 0, EDGE_AUX);
            else
                __builtin_amdgcn_global_load_lds(rec_bytes + ((uint64_t)rec << 4), (lds_void_ptr)(ring + (uint32_t)i * 1024u), 16, 0, AUX);
        }
    }
}

// The workgroup's 256 * R units -> R tiles per wave, longest units first.  beg / cnt: first item and item count of
// the thread's R units (local index j * 256 + tid; `n_here` of them exist).  info[r] = {first item, items, local index or kPadUnit, 0} of
// this lane's unit in the wave's r-th tile.  The sort scratch lives in the rings: call before any streaming;
// the rings are free again when this returns (it ends with a barrier, which also covers the caller's table staging).
template <int R>
__device__ __forceinline__ void wg_sort_into_tiles(unsigned char* rings, const uint32_t (&beg)[R], const uint32_t (&cnt)[R],
                                                   const uint32_t n_here, const uint32_t tid,
                                                   const uint32_t lane, const uint32_t wave, uint4 (&info)[R])
{
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(rings);      // kMaxSortKey + 1 buckets
    uint32_t* s_start = s_hist + (kMaxSortKey + 1);
    uint32_t* s_wsum = s_start + (kMaxSortKey + 1);              // kWavesPerBlock
    uint4* s_info = reinterpret_cast<uint4*>(s_wsum + 8);        // per sorted position
    uint32_t key[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t nblk = cnt[j] ? ((beg[j] & 7u) + cnt[j] + 7u) >> 3 : 0u;
        key[j] = min(nblk, kMaxSortKey);
    }
    for (uint32_t i = tid; i <= kMaxSortKey; i += kBlock) s_hist[i] = 0u;
    __syncthreads();
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
        const uint32_t local = (uint32_t)j * kBlock + tid;
        s_info[s_start[key[j]] + rank[j]] = make_uint4(beg[j], cnt[j], local < n_here ? local : kPadUnit, 0u);
    }
    __syncthreads();
    // the r-th tile of this wave in snake order
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t tile = (uint32_t)r * kWavesPerBlock + ((r & 1) ? (uint32_t)kWavesPerBlock - 1u - wave : wave);
        info[r] = s_info[tile * kWave + lane];
    }
    __syncthreads();   // the rings are free from here on
}

// the block count of the longest and of the shortest unit of a sorted tile (wave-uniform)
__device__ __forceinline__ void tile_block_range(const uint32_t nblk, uint32_t& max_blk, uint32_t& min_blk)
{
    // sorted longest first: the tile's first lane has the most blocks -- unless it sits in the last sort bucket,
    // which holds every longer unit in arrival order
    max_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)nblk);
    if (max_blk >= kMaxSortKey) {
        uint32_t m = nblk;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, kWave));
        max_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
    }
    uint32_t m = nblk;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) m = min(m, (uint32_t)__shfl_xor((int)m, d, kWave));
    min_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
}

// Result records leave with non-temporal stores: a result line is written once and read by nobody on the device
// (measured on the 1 M-unit pass: nt 0.340 ms, sc1 0.354, default policy 0.359).  The store is the compiler's own
// (__builtin_nontemporal_store -> global_store_dwordx4 ... nt): an inline-assembly store is invisible to the
// hazard recognizer -- a 16-byte VMEM store whose data registers are overwritten by the very next instruction needs a
// wait state -- and one build of the two-tile window kernel, whose register allocation put a spill reload right
// behind such a store, wrote a foreign dword into four pieces of a few result records.
// SVT_STORE_POLICY (measurements only): "" / " sc1" / " sc0 sc1" select an assembly store with that policy (+ s_nop).
#ifndef SVT_STORE_DIRECT
#define SVT_STORE_DIRECT 0    // 1: every lane stores the eight pieces of its own record (no LDS staging)
#endif

__device__ __forceinline__ void store_piece(uint4* dst, const uint4 v)
{
    const u32x4 vv = {v.x, v.y, v.z, v.w};
#ifdef SVT_STORE_POLICY
    asm volatile("global_store_dwordx4 %0, %1, off" SVT_STORE_POLICY "\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
#else
    __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(dst));
#endif
}

// result records of a tile: lane-major into the ring, unit-major out of it, one full 128-byte line per eight lanes
__device__ __forceinline__ void store_results_through_ring(unsigned char* ring, const uint4 (&piece)[8], const uint32_t unit,
                                                           const uint32_t lane, svt_result* __restrict__ out)
{
#if SVT_STORE_DIRECT
    if (unit != kPadUnit) {
#pragma unroll
        for (int p = 0; p < 8; ++p) store_piece(reinterpret_cast<uint4*>(out + unit) + p, piece[p]);
    }
    return;
#endif
    const uint32_t o = lane >> 3, rr = lane & 7u, sw = (lane >> 1) & 7u;
    const uint32_t col_even = (rr ^ (o >> 1)) << 4, col_odd = col_even ^ 64u;
    uint4* st = reinterpret_cast<uint4*>(ring + lane * 128u);
#pragma unroll
    for (int p = 0; p < 8; ++p) st[(uint32_t)p ^ sw] = piece[p];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t dst_unit = (uint32_t)__shfl((int)unit, 8 * i + (int)o, kWave);
        const uint4 v = *reinterpret_cast<const uint4*>(ring + (uint32_t)i * 1024u + o * 128u + ((i & 1) ? col_odd : col_even));
        if (dst_unit != kPadUnit) store_piece(reinterpret_cast<uint4*>(out + dst_unit) + rr, v);
    }
}

#ifndef SVT_STORE_UNROLL
#define SVT_STORE_UNROLL 2
#endif
#ifndef SVT_STORE_BARRIER
#define SVT_STORE_BARRIER 0
#endif
// One routine for both result record forms -- 128-byte svt_result (P = 8 sixteen-byte pieces) and the 96-byte svt_result96 of
// SVT_FLAG_RESULT96 (P = 6: GL, SQ, the five tallies, QR / QA / GQ, GT; the other counts follow from the tallies on the host,
// svt_results_expand96) -- with P a wave-uniform run-time value: two specialised routines behind a branch made the compiler
// hoist their common parts in front of it and cost the kernel 30 VGPRs.  The P pieces of a unit leave through P consecutive
// lanes, 64 units in P store instructions; a 96-byte record starts on a 32-byte boundary, so whole 32-byte sectors are written.
__device__ __forceinline__ void store_result_records_through_ring(unsigned char* ring, const uint4 (&piece)[8], const uint32_t unit,
                                                                  const uint32_t lane, unsigned char* __restrict__ out, const uint32_t P,
                                                                  const uint32_t sorted_base = 0xFFFFFFFFu)
{
    const uint32_t sw = (lane >> 1) & 7u;
    uint4* st = reinterpret_cast<uint4*>(ring + lane * 128u);
#pragma unroll
    for (int p = 0; p < 8; ++p) st[(uint32_t)p ^ sw] = piece[p];
    const uint32_t stride = P * 16u;
#if SVT_STORE_UNROLL == 8
#pragma unroll
#elif SVT_STORE_UNROLL == 1
#pragma unroll 1
#else
#pragma unroll 2
#endif
    for (int i = 0; i < 8; ++i) {
        if ((uint32_t)i >= P) break;                       // (wave-uniform)
        if (SVT_STORE_BARRIER) __builtin_amdgcn_sched_barrier(0);
        const uint32_t idx = (uint32_t)i * kWave + lane, u = P == 8u ? idx >> 3 : idx / 6u, p = idx - u * P;
        const uint32_t dst_unit = (uint32_t)__shfl((int)unit, (int)u, kWave);
        const uint4 v = *reinterpret_cast<const uint4*>(ring + u * 128u + ((p ^ ((u >> 1) & 7u)) << 4));
        if (sorted_base != 0xFFFFFFFFu) {
            // the tile's 64 records in the tile's own (length-sorted) order: one store instruction writes 1 KB of whole lines
            store_piece(reinterpret_cast<uint4*>(out + (uint64_t)sorted_base * stride) + idx, v);
            continue;
        }
        if (dst_unit != kPadUnit) store_piece(reinterpret_cast<uint4*>(out + (uint64_t)dst_unit * stride) + p, v);
    }
}

}  // namespace svt

#endif  // SVT_RING_ENGINE_H
