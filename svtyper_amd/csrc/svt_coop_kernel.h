// svt_coop_kernel.h -- the genotype pass for launches of LESS THAN ONE ROUND of resident workgroups
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
//
// svt_stream_kernel gives a unit to a lane, and the lane does everything a record needs: a dozen LDS look-ups in two
// dependent round trips, the products, the six additions.  With the chip full of such waves that is a memory-bound
// stream.  With a quarter of the chip's wave slots taken -- an 8-GPU shard of configs[3], a driver's chunk of
// a few thousand breakpoints -- it is a latency chain: a wave that has its SIMD to itself needs ~1.5 us per 128-byte block,
// the workgroup's longest tile has ~22 of them, and the pass takes 43 us however few the units (profiles/r05_small_launch_probes.txt).
//
// Only the SUMS have to follow the reference's order (classic.py:296 `sorted(query_name)`, :306-405 the `+=`;
// singlesample.py:367-378).  What is added -- prob_mapq look-ups, the p_concordant rank compare, pmA * pmB * w -- depends on
// the record alone.  So a workgroup here is five waves with two jobs:
//
//   * eight PRODUCER waves: per step, the 64 units x 8 records of one tile step are 512 independent items, one per lane.
//     A lane loads its record straight from HBM into registers (eight lanes of a unit read one 128-byte line; four steps
//     ahead, no LDS ring), does the look-ups of svt_stream_kernel's record_single and leaves the six addends of the record
//     {rs_a, rs_b | p_seq, p_clip | pmA pmB w_alt, pmA pmB w_ref} in an LDS stage;
//   * three CONSUMER waves, one per pair of addends: one unit per lane as before; per step a consumer reads its plane of its
//     unit's eight records and adds in record order -- the same operands, the same operations, the same order per accumulator
//     as record_single: the same bits.  (The accumulators are independent of each other, so they can live in different waves.)
//
// One barrier per step hands a stage from the producers to the consumers (two stages).  A producer lane's work does not
// depend on its unit's length, a consumer's chain per step is 16 additions, and a wave that has a SIMD nearly to itself issues
// an instruction every ~8 cycles whatever it is: what counts is the instructions on the longest wave, ~60 per step here
// against ~425 per block there.  The workgroup's 64-unit tiles (sorted by length as in svt_ring_engine.h) follow each other
// without draining the producers' loads; the tallies of a finished tile wait in LDS and the epilogues of the four tiles run on
// four waves at the end.
#ifndef SVT_COOP_KERNEL_H
#define SVT_COOP_KERNEL_H

#include "svt_stream_kernel.h"

#ifndef SVT_COOP_PRODUCERS
#define SVT_COOP_PRODUCERS 8
#endif
#ifndef SVT_COOP_DEPTH
#define SVT_COOP_DEPTH 4      // tile steps of records a producer lane keeps in flight (HBM latency ~0.45 us, a step ~0.2-0.3 us)
#endif
#ifndef SVT_COOP_PROBE
#define SVT_COOP_PROBE 0      // timing only (wrong results), bits: 1 = the producers do not write their addends, 2 = the consumer sums nothing,
                              // 4 = no record loads, 8 = no look-ups
#endif
#ifndef SVT_COOP_WAVES_PER_SIMD
#define SVT_COOP_WAVES_PER_SIMD 6   // the register allocation must allow this many waves per SIMD
#endif
#ifndef SVT_COOP_TRACE
#define SVT_COOP_TRACE 0      // debugging: workgroup 0 prints the shader-clock time of its phases
#endif
#ifndef SVT_COOP_NT
#define SVT_COOP_NT 1         // record loads non-temporal
#endif

namespace svt {

constexpr int kCoopProducers = SVT_COOP_PRODUCERS;
constexpr int kCoopConsumers = 3;                         // one per addend plane: {ref_seq} {alt_seq, alt_clip} {alt_span, ref_span}
constexpr int kCoopWaves = kCoopConsumers + kCoopProducers;
constexpr int kCoopBlock = kCoopWaves * kWave;
constexpr int kCoopItems = 8 / kCoopProducers;            // records of a tile step per producer lane: 512 / (64 * producers)
static_assert(kCoopItems * kCoopProducers == 8, "the producers share a tile step's 512 records evenly");
static_assert(kCoopWaves >= kBlock / kWave, "four waves sort the units and run the epilogues");
constexpr int kCoopDepth = SVT_COOP_DEPTH;
static_assert(kCoopDepth % 2 == 0, "the stage parity of an unrolled step must be static");
constexpr uint32_t kCoopStageBytes = 8u * 3u * 64u * 16u; // one tile step of addends: [slot 8][plane 3][position 64] x 16 bytes
constexpr uint32_t kCoopTiles = kBlock / kWave;           // 64-unit tiles of a workgroup's (up to) 256 units
// the cooperative region of the workgroup's LDS, byte offsets from StreamArgs::lds_rings (128-byte aligned)
constexpr uint32_t kCoopTileAt = 0;                                  // uint4[256]  {first record, records, sub2, flags} by sorted position
constexpr uint32_t kCoopUnitAt = kCoopTileAt + kBlock * 16u;         // uint32[256] unit index (kPadUnit: none)
constexpr uint32_t kCoopMiscAt = kCoopUnitAt + kBlock * 4u;          // uint32 tmax[4], cont[4]
constexpr uint32_t kCoopTallyAt = kCoopMiscAt + 128u;                // double[5][256] tallies of the finished tiles
constexpr uint32_t kCoopStageAt = kCoopTallyAt + 5u * kBlock * 8u;   // two addend stages; before the steps: sort scratch; after them: result rings
constexpr uint32_t kCoopRegionBytes = kCoopStageAt + 2u * kCoopStageBytes;
static_assert(kCoopStageAt % 128u == 0u, "stages (and the result rings that reuse them) are line-aligned");
static_assert(2u * kCoopStageBytes >= kCoopTiles * kRingBytes + 3u * (kMaxSortKey + 1u) * 4u, "the result rings / the sort scratch fit the stages");
// flags word of a tile entry
constexpr uint32_t kCoopFmask = 7u, kCoopDel16 = 16u, kCoopSvtypeShift = 8u, kCoopUflagsShift = 16u;

// ---- the record arithmetic of record_single (svt_stream_kernel.h) in two halves ---------------------------------------
struct CoopLook {
    double pm_a, pm_b, rs_a, rs_b, s0, s1, c0, c1;
    int32_t thr1;
    uint32_t h2;
};
// every table look-up that depends on the record alone (classic.py:306-328, 339-358; parsers.py:861-882)
__device__ __forceinline__ void coop_look(const u32x4 w, const uint32_t kmin, const uint32_t nb, const uint32_t sub2, const uint32_t hist_at, CoopLook& L)
{
    const uint32_t i1 = min(w.x - kmin, nb), i2 = min(w.x - sub2, nb);
    L.thr1 = lds_i16(kSBins + (i1 << 1));
    L.h2 = lds_u16(hist_at + (i2 << 1));
    L.pm_a = lds_f64(kSPm + byte0_x8(w.y));
    L.pm_b = lds_f64(kSPm + byte1_x8(w.y));
    L.rs_a = lds_f64(kSPm + byte2_x8(w.y));
    L.rs_b = lds_f64(kSPm + byte3_x8(w.y));
    L.s0 = lds_f64(kSPmHalf + byte0_x8(w.z));
    L.s1 = lds_f64(kSPmHalf + byte1_x8(w.z));
    L.c0 = lds_f64(kSPmHalf + byte2_x8(w.z));
    L.c1 = lds_f64(kSPmHalf + byte3_x8(w.z));
}

typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4_;

__device__ __forceinline__ u32x4 pack2d_v(const double x, const double y)
{
    const uint4 v = pack2d(x, y);
    return u32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ double lo_f64(const u32x4 v) { return __hiloint2double((int)v.y, (int)v.x); }
__device__ __forceinline__ double hi_f64(const u32x4 v) { return __hiloint2double((int)v.w, (int)v.z); }
__device__ __forceinline__ double lo_abs_f64(const u32x4 v) { return __hiloint2double((int)(v.y & 0x7fffffffu), (int)v.x); }

// the cooperative arguments ride in StreamArgs (svt_stream_kernel.h): lds_rings = the cooperative region, units_per_wg <= 256

template <bool SSO, int MODE>
__global__ __launch_bounds__(kCoopBlock, SVT_COOP_WAVES_PER_SIMD) void svt_coop_kernel(const StreamArgs a)
{
    static_assert(MODE == kSingleLds, "one library (library windows: next)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();   // tables at absolute LDS addresses
    const uint64_t trace_t0 = SVT_COOP_TRACE ? clock64() : 0;
    const uint32_t tid = threadIdx.x, lane = tid % kWave;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));
    unsigned char* region = smem + a.lds_rings;
    uint4* s_tile = reinterpret_cast<uint4*>(region + kCoopTileAt);
    uint32_t* s_unit = reinterpret_cast<uint32_t*>(region + kCoopUnitAt);
    uint32_t* s_tmax = reinterpret_cast<uint32_t*>(region + kCoopMiscAt);
    uint32_t* s_cont = s_tmax + 4;
    double* s_tally = reinterpret_cast<double*>(region + kCoopTallyAt);
    unsigned char* stages = region + kCoopStageAt;
    const uint32_t stage_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)stages;

    const uint32_t wg_base = a.unit_begin + blockIdx.x * a.units_per_wg;
    const uint32_t n_here = min(a.units_per_wg, a.unit_end - wg_base);
    const bool sorter = tid < (uint32_t)kBlock;    // the first four waves hold the workgroup's (up to) 256 units

    // ---- this thread's unit: record range and header (the loads overlap the table staging below)
    uint32_t beg = 0u, cnt = 0u;
    svt_unit U{};
    if (sorter && tid < n_here) {
        const uint64_t lo = a.rec_offset[wg_base + tid], hi = a.rec_offset[wg_base + tid + 1];
        beg = (uint32_t)lo;
        cnt = (uint32_t)(hi - lo);
        U = a.units[wg_base + tid];
    }
    // ---- tables (the layout of svt_stream_kernel: kSPm ...)
    for (uint32_t i = tid; i < 256; i += kCoopBlock) {
        const double p = a.pm[i];
        reinterpret_cast<double*>(smem + kSPm)[i] = p;
        reinterpret_cast<double*>(smem + kSPmHalf)[i] = p * 0.5;
    }
    if (tid < 32) {
        const PairWeights pw = a.wtab[tid];
        reinterpret_cast<double*>(smem + kSWtab)[tid] = pw.w_alt;
        reinterpret_cast<double*>(smem + kSWtab + kSWref)[tid] = pw.w_ref;
    }
    {
        int16_t* s_thr = reinterpret_cast<int16_t*>(smem + kSBins);
        uint16_t* s_hst = reinterpret_cast<uint16_t*>(smem + kSBins) + a.total_bins;
        for (uint32_t i = tid; i < a.total_bins; i += kCoopBlock) {
            const Bin bn = a.bins[i];
            s_thr[i] = (int16_t)bn.thr;
            s_hst[i] = (uint16_t)bn.hist;
        }
    }
    if (a.l10_where == kL10Shared) {
        double* s_l10 = reinterpret_cast<double*>(smem + a.lds_l10);
        for (uint32_t i = tid; i < a.n_l10; i += kCoopBlock) s_l10[i] = a.l10[i];
    }
    for (uint32_t i = tid; i < 5u * kBlock; i += kCoopBlock) s_tally[i] = 0.0;
    if (tid < 8) s_tmax[tid] = 0u;     // tmax[4], cont[4]

    // ---- counting sort of the units by block count, longest first (wg_sort_into_tiles, with the other waves only meeting the
    // barriers).  What moves is the tile entry: everything a record's look-ups need to know of its unit (record_single's StreamCtx).
    {
        uint32_t* s_hist = reinterpret_cast<uint32_t*>(stages);
        uint32_t* s_start = s_hist + (kMaxSortKey + 1);
        uint32_t* s_wsum = s_start + (kMaxSortKey + 1);
        const uint32_t nblk = cnt ? ((beg & 7u) + cnt + 7u) >> 3 : 0u;
        // (a unit without records still sorts in front of the padding threads: the workgroup's units fill its first tiles)
        const uint32_t key = sorter && tid < n_here ? min(nblk + 1u, kMaxSortKey) : 0u;
        if (sorter) s_hist[tid] = 0u;
        __syncthreads();
        uint32_t rank = 0u;
        if (sorter) rank = atomicAdd(&s_hist[key], 1u);
        __syncthreads();
        uint32_t h = 0u, incl = 0u;
        if (sorter) {
            h = s_hist[kMaxSortKey - tid];
            incl = wave_inclusive_scan(h, lane);
            if (lane == kWave - 1) s_wsum[wave] = incl;
        }
        __syncthreads();
        if (sorter) {
            uint32_t before = 0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) before += (uint32_t)w < wave ? s_wsum[w] : 0u;
            const uint32_t pos = before + incl - h;      // first sorted position of bucket kMaxSortKey - tid
            s_start[kMaxSortKey - tid] = pos;
        }
        __syncthreads();
        if (sorter) {
            const bool is_del = U.svtype == SVT_SVTYPE_DEL;
            const bool small_del = is_del && ((double)U.pos_delta < a.lib0.sd2);   // classic.py:339,383
            const uint32_t flags = (small_del ? 0u : kCoopFmask) | (is_del ? kCoopDel16 : 0u) | ((uint32_t)U.svtype << kCoopSvtypeShift) |
                                   ((uint32_t)U.flags << kCoopUflagsShift);
            const uint32_t sub2 = is_del ? (uint32_t)U.var_length + (uint32_t)a.lib0.key_min : 0x80000000u;
            const uint32_t pos = s_start[key] + rank;
            s_tile[pos] = make_uint4(beg, cnt, sub2, flags);
            s_unit[pos] = tid < n_here ? wg_base + tid : kPadUnit;
            // the tile's longest unit (sorted: its first -- unless it sits in the last bucket, which keeps arrival order)
            atomicMax(&s_tmax[pos >> 6], nblk);
        }
        __syncthreads();
    }
    uint32_t tmax[kCoopTiles];
    uint32_t steps = 0u;
#pragma unroll
    for (uint32_t t = 0; t < kCoopTiles; ++t) {
        tmax[t] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tmax[t]);
        steps += tmax[t];
    }
    // (sorted longest first: an empty tile is followed by empty tiles only)
    const uint32_t rounds = (steps + 1u + kCoopDepth - 1u) / kCoopDepth;   // the consumers trail the producers by one step
    const uint64_t trace_t1 = SVT_COOP_TRACE ? clock64() : 0;
    RecordCheck<MODE> check;
    if (wave >= (uint32_t)kCoopConsumers) {
        // =================================================== producers ===================================================
        const uint32_t p = wave - (uint32_t)kCoopConsumers;
        const uint32_t slot = lane & 7u;
        const char* rec_bytes = reinterpret_cast<const char*>(a.records);
        const uint32_t kmin = (uint32_t)a.lib0.key_min, nb = a.lib0.n_bins, hist_at = kSBins + a.total_bins * 2u;
        const uint32_t last_rec = a.last_blk * kBlockRecords + 7u;
        uint32_t upos[kCoopItems], st_at[kCoopItems];     // the item's unit inside its tile; where its addends go inside a stage
#pragma unroll
        for (int j = 0; j < kCoopItems; ++j) {
            upos[j] = ((uint32_t)j * kCoopProducers + p) * 8u + (lane >> 3);
            st_at[j] = (slot * 3u * 64u + (upos[j] ^ (slot << 1))) * 16u;
        }
        // load cursor (kCoopDepth steps ahead) and work cursor: tile, steps left in it, and the tile's entries of this lane's items
        uint32_t tl = 0u, left_l = tmax[0], tw = 0u, left_w = tmax[0];
        uint32_t l_base[kCoopItems], l_first[kCoopItems], l_nrec[kCoopItems];     // l_base: the record this lane loads at the load cursor
        uint32_t w_base[kCoopItems], w_first[kCoopItems], w_nrec[kCoopItems], w_sub2[kCoopItems], w_flags[kCoopItems];
        auto entries_for_loads = [&]() {
#pragma unroll
            for (int j = 0; j < kCoopItems; ++j) {
                const uint4 e = s_tile[min(tl, kCoopTiles - 1u) * kWave + upos[j]];
                l_first[j] = e.x;
                l_nrec[j] = e.y;
                l_base[j] = (e.x & ~7u) + slot;
            }
        };
        auto entries_for_work = [&]() {
#pragma unroll
            for (int j = 0; j < kCoopItems; ++j) {
                const uint4 e = s_tile[min(tw, kCoopTiles - 1u) * kWave + upos[j]];
                w_first[j] = e.x;
                w_nrec[j] = e.y;
                w_sub2[j] = e.z;
                w_flags[j] = e.w;
                w_base[j] = (e.x & ~7u) + slot;
            }
        };
        entries_for_loads();
        entries_for_work();
        u32x4 buf[kCoopDepth][kCoopItems];
        // Every lane loads at every step -- an item outside its unit's records (the neighbours' records in the unit's first and
        // last line, steps past the end of a shorter unit, steps past the end of the workgroup) reads its unit's first line
        // again and is replaced by the neutral record when its turn comes: straight-line code whose load buffers keep their
        // registers (a conditional load made the compiler copy the buffers around behind a full vmcnt(0)).
        auto issue_loads = [&](u32x4 (&dst)[kCoopItems]) {
#pragma unroll
            for (int j = 0; j < kCoopItems; ++j) {
                const bool valid = l_base[j] - l_first[j] < l_nrec[j];   // first <= record < first + records, unsigned
                const uint32_t at = min(valid ? l_base[j] : l_first[j], last_rec);
                const u32x4* src = reinterpret_cast<const u32x4*>(rec_bytes + ((uint64_t)at << 4));
                if (SVT_COOP_PROBE & 4) dst[j] = u32x4{at, at, at, 0u};
                else dst[j] = SVT_COOP_NT ? __builtin_nontemporal_load(src) : *src;
                l_base[j] += kBlockRecords;
            }
            if (left_l && --left_l == 0u) {
                ++tl;
                left_l = tl < kCoopTiles ? s_tmax[min(tl, kCoopTiles - 1u)] : 0u;
                left_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)left_l);
                entries_for_loads();
            }
        };
#pragma unroll
        for (int d = 0; d < kCoopDepth; ++d) issue_loads(buf[d]);

        // (whole rounds of kCoopDepth steps; the steps past the end sum nothing: their items are neutral and nobody reads their stage)
        for (uint32_t r = 0; r < rounds; ++r) {
#pragma unroll
            for (int d = 0; d < kCoopDepth; ++d) {
                const uint32_t st = stage_addr + ((uint32_t)d & 1u) * kCoopStageBytes;
                bool any_cont = false;
#pragma unroll
                for (int j = 0; j < kCoopItems; ++j) {
                    const bool mine = left_w != 0u && w_base[j] - w_first[j] < w_nrec[j];
                    w_base[j] += kBlockRecords;
                    u32x4 w = buf[d][j];
                    // the neutral record: MAPQ 0 everywhere adds +0.0 to every sum (svt_stream_kernel.h), no flag, span 0
                    w.x = mine ? w.x : 0u;
                    w.y = mine ? w.y : 0u;
                    w.z = mine ? w.z : 0u;
                    w.w = mine ? w.w : 0u;
                    check.see(w);
                    CoopLook L;
                    if (SVT_COOP_PROBE & 8) {
                        L.pm_a = L.pm_b = L.rs_a = L.rs_b = L.s0 = L.s1 = L.c0 = L.c1 = __hiloint2double((int)w.x, (int)w.y);
                        L.thr1 = (int32_t)w.z;
                        L.h2 = w.w;
                    } else
                    coop_look(w, kmin, nb, w_sub2[j], hist_at, L);
                    const bool p_conc = (int32_t)L.h2 <= L.thr1;
                    const uint32_t wt0 = kSWtab + (w_flags[j] & kCoopDel16) * 8u;
                    const uint32_t wa = (p_conc ? wt0 + 8u * 8u : wt0) | ((w.w & w_flags[j] & kCoopFmask) << 3);   // &w_alt[f3 | p_conc << 3 | del16]
                    const double w_alt = lds_f64(wa), w_ref = lds_f64(wa + kSWref);
                    const double pp = L.pm_a * L.pm_b;
                    double rs_a = L.rs_a, p_seq = L.s0 + L.s1;
                    if (SSO) {   // a continuation record (singlesample.py:246-276: the fragment's sums go on) travels as the sign of two addends >= +0.0
                        const bool cont = (w.w & SVT_REC_CONTINUATION) != 0u;
                        any_cont = any_cont || cont;
                        rs_a = cont ? -rs_a : rs_a;
                        p_seq = cont ? -p_seq : p_seq;
                    }
                    const uint32_t at = st + st_at[j];
                    if (SVT_COOP_PROBE & 1) {
                        if (rs_a + L.rs_b + L.s0 + L.s1 + L.c0 + L.c1 + pp * w_alt + pp * w_ref == 1.2345e-300) *reinterpret_cast<lds_u32x4*>((size_t)at) = pack2d_v(rs_a, L.rs_b);
                        continue;
                    }
                    *reinterpret_cast<lds_u32x4*>((size_t)at) = pack2d_v(rs_a, L.rs_b);
                    *reinterpret_cast<lds_u32x4*>((size_t)(at + 1024u)) = pack2d_v(p_seq, L.c0 + L.c1);
                    *reinterpret_cast<lds_u32x4*>((size_t)(at + 2048u)) = pack2d_v(pp * w_alt, pp * w_ref);
                }
                if (SSO && __any(any_cont) && lane == 0u) atomicOr(&s_cont[(r * kCoopDepth + (uint32_t)d) & 3u], 1u);
                if (left_w && --left_w == 0u) {
                    ++tw;
                    left_w = tw < kCoopTiles ? s_tmax[min(tw, kCoopTiles - 1u)] : 0u;
                    left_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)left_w);
                    entries_for_work();
                }
                issue_loads(buf[d]);
                __syncthreads();
            }
        }
    } else {
        // =================================================== consumers ===================================================
        // wave c sums plane c of the addends: {rs_a, rs_b} -> ref_seq | {p_seq, p_clip} -> alt_seq, alt_clip | {alt, ref} -> alt_span, ref_span
        uint32_t rd[8];   // where this lane's unit keeps this plane's addends of record `slot` inside a stage
#pragma unroll
        for (uint32_t s = 0; s < 8u; ++s) rd[s] = stage_addr + ((s * 3u + wave) * 64u + (lane ^ (s << 1))) * 16u;
        uint32_t tc = 0u, left_c = tmax[0];
        double acc0 = 0.0, acc1 = 0.0, loc0 = 0.0, loc1 = 0.0;   // two site sums; sso: their fragment-local sums
        for (uint32_t g = 0; g < rounds * kCoopDepth; ++g) {
            if (g >= 1u && left_c != 0u) {
                const uint32_t off = ((g - 1u) & 1u) * kCoopStageBytes;
                u32x4 v[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) v[s] = *reinterpret_cast<lds_cu32x4_*>((size_t)(rd[s] + off));
                const bool has_cont = SSO && wave != 2u && s_cont[(g - 1u) & 3u] != 0u;
                if (SVT_COOP_PROBE & 2) {
                } else if (wave == 2u) {   // classic.py:339-405 / singlesample.py:278-353: the spans know no fragments
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        acc0 += lo_f64(v[s]);
                        acc1 += hi_f64(v[s]);
                    }
                } else if (!SSO) {
                    if (wave == 0u) {      // classic.py:306-315
#pragma unroll
                        for (int s = 0; s < 8; ++s) acc0 = (acc0 + lo_f64(v[s])) + hi_f64(v[s]);
                    } else {               // classic.py:316-328
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            acc0 += lo_f64(v[s]);
                            acc1 += hi_f64(v[s]);
                        }
                    }
                } else if (!has_cont) {    // no record of this step continues a fragment: record_weights<SSO, ., false>
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        acc0 += loc0;
                        if (wave == 0u) {
                            loc0 = lo_f64(v[s]) + hi_f64(v[s]);
                        } else {
                            acc1 += loc1;
                            loc0 = lo_f64(v[s]);
                            loc1 = hi_f64(v[s]);
                        }
                    }
                } else {                   // singlesample.py:246-276,367-372 -- the general form of record_weights<SSO>
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const bool cont = (int32_t)v[s].y < 0;
                        const double x = lo_abs_f64(v[s]), y = hi_f64(v[s]);
                        acc0 += cont ? 0.0 : loc0;
                        if (wave == 0u) {
                            loc0 = ((cont ? loc0 : 0.0) + x) + y;
                        } else {
                            acc1 += cont ? 0.0 : loc1;
                            loc0 = (cont ? loc0 : 0.0) + x;
                            loc1 = (cont ? loc1 : 0.0) + y;
                        }
                    }
                }
                if (--left_c == 0u) {
                    // the tile is summed: its tallies wait in LDS for the epilogues
                    if (SSO && wave != 2u) {  // flush the last fragment (singlesample.py:370-372)
                        acc0 += loc0;
                        acc1 += loc1;
                    }
                    const uint32_t pos = tc * kWave + lane;
                    // s_tally planes: ref_seq, alt_seq, alt_clip, ref_span, alt_span
                    if (wave == 0u) s_tally[pos] = acc0;
                    else if (wave == 1u) { s_tally[kBlock + pos] = acc0; s_tally[2 * kBlock + pos] = acc1; }
                    else { s_tally[4 * kBlock + pos] = acc0; s_tally[3 * kBlock + pos] = acc1; }
                    acc0 = acc1 = loc0 = loc1 = 0.0;
                    ++tc;
                    left_c = tc < kCoopTiles ? s_tmax[min(tc, kCoopTiles - 1u)] : 0u;
                    left_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)left_c);
                }
            }
            // (four flags in turn: the one the consumers read a step ago is cleared two steps before the producers raise it again)
            if (SSO && g >= 2u && wave == 0u && lane == 0u) s_cont[(g - 2u) & 3u] = 0u;
            __syncthreads();
        }
    }
    const uint64_t trace_t2 = SVT_COOP_TRACE ? clock64() : 0;
    __syncthreads();   // every tile's tallies are in LDS; the stages are free for the result rings

    // ---- epilogues: wave w takes tile w (classic.py:425-513 in unit_epilogue)
    const uint32_t n_tiles = (a.units_per_wg + kWave - 1u) / kWave;   // tiles this workgroup owns result slots for (uniform over the launch)
    if (wave < n_tiles && wave < kCoopTiles) {
        const uint32_t pos = wave * kWave + lane;
        Acc acc = {s_tally[pos], s_tally[kBlock + pos], s_tally[2 * kBlock + pos], s_tally[3 * kBlock + pos], s_tally[4 * kBlock + pos], 0.0, 0.0, 0.0};
        if (SVT_COOP_PROBE) {   // timing only: the sums are garbage and must not index the log10 table -- mask them with what the compiler cannot fold
            const uint64_t keep = a.unit_end < a.unit_begin ? ~0ull : 0ull;
            acc.ref_seq = __longlong_as_double((long long)((uint64_t)__double_as_longlong(acc.ref_seq) & keep));
            acc.alt_seq = __longlong_as_double((long long)((uint64_t)__double_as_longlong(acc.alt_seq) & keep));
            acc.alt_clip = __longlong_as_double((long long)((uint64_t)__double_as_longlong(acc.alt_clip) & keep));
            acc.ref_span = __longlong_as_double((long long)((uint64_t)__double_as_longlong(acc.ref_span) & keep));
            acc.alt_span = __longlong_as_double((long long)((uint64_t)__double_as_longlong(acc.alt_span) & keep));
        }
        const uint32_t unit = s_unit[pos], flags = s_tile[pos].w;
        const double* lds_l10 = reinterpret_cast<const double*>(smem + a.lds_l10);
        uint4 piece[8];
        unit_epilogue(acc, (flags >> kCoopSvtypeShift) & 0xffu, flags >> kCoopUflagsShift, a.c, lds_l10, a.l10, a.l10_where == kL10Shared ? a.l10_lds_entries : 0u, piece);
        uint32_t unit_out = unit;
        if (a.out_samples > 1u && unit != kPadUnit) {
            const uint32_t sample = unit / a.out_sites;
            unit_out = (unit - sample * a.out_sites) * a.out_samples + sample;
        }
        uint32_t tile_slot = 0xFFFFFFFFu;
        if (a.result96) {
            piece[5] = make_uint4(piece[5].x, piece[7].y, unit_out, 0u);
            tile_slot = a.slot_begin + (blockIdx.x * n_tiles + wave) * kWave;
        }
        store_result_records_through_ring(stages + wave * kRingBytes, piece, unit_out, lane, reinterpret_cast<unsigned char*>(a.out), a.result96 ? 6u : 8u, tile_slot);
    }
    const uint32_t bad = check.bits(a.n_libs);
    if (bad) atomicOr(a.err, bad);
    if (SVT_COOP_TRACE && (blockIdx.x % 61u) == 0 && lane == 0 && (wave == 0 || wave == 5)) {
        const uint64_t trace_t3 = clock64();
        printf("coop wg %u wave %u (realtime %llu): steps %u (tiles %u %u %u %u) prologue %llu loop %llu (%.0f per step) epilogue %llu cycles\n", blockIdx.x, wave, (unsigned long long)wall_clock64(), steps, tmax[0], tmax[1], tmax[2], tmax[3],
               (unsigned long long)(trace_t1 - trace_t0), (unsigned long long)(trace_t2 - trace_t1), (double)(trace_t2 - trace_t1) / (double)max(steps, 1u),
               (unsigned long long)(trace_t3 - trace_t2));
    }
}

}  // namespace svt

#endif  // SVT_COOP_KERNEL_H
