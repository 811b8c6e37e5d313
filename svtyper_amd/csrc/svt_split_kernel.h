// svt_split_kernel.h -- K lanes per unit for launches of less than one round of resident workgroups
// Internal header of libsvtyper_hip.so (translation unit: svt_small_kernels.hip).
//
// One lane per unit makes a full launch a memory-bound stream (svt_stream_kernel.h) -- and a partial launch a latency chain:
// a wave that shares its SIMD with at most one other issues a dependent instruction every ~8 cycles, the workgroup's longest
// tile has ~22 blocks of ~425 instructions, and nothing shortens that chain but fewer instructions on the wave that walks it
// (profiles/r05_small_launch_probes.txt).  Only the SUMS have to follow the reference's order (classic.py:296
// `sorted(query_name)`, :306-405 the `+=`; singlesample.py:367-378); what is added depends on the record alone.
//
// Here a unit belongs to K adjacent lanes (K = 2 or 4).  Lane h of the group takes records h * 8 / K ... of every 128-byte
// block: its look-ups and products are a K-th of a block's, and the sums travel through the group in record order -- lane 0
// adds its records to the running sums, hands them to lane 1 (one DPP row shift per register), ... lane K - 1 hands them back
// to lane 0 for the next block.  Same operands, same operations, same order per accumulator as record_single: the same bits.
// A wave walks 64 / K units; per block it issues (8 / K) x 37 instructions of look-ups + 48 additions + 10 K register shifts
// instead of 8 x 45: the chain per block is 0.66 (K = 2) or 0.5 (K = 4) of the one-lane kernel's, for 1.3 x / 2 x its
// instructions in total -- which a partial launch has the issue slots for.
//
// The rest is the streaming kernel's: the workgroup's 256 units sorted by length into tiles (64 / K units per wave, 4 K waves),
// LDS-DMA of one line per unit and step into a per-wave ring, the neutral record for slots outside a unit; the tallies of the
// K-lane groups go through LDS to four waves that run the epilogues one unit per lane, so the result records are those of
// the streaming kernel, slot for slot.
#ifndef SVT_SPLIT_KERNEL_H
#define SVT_SPLIT_KERNEL_H

#include "svt_stream_kernel.h"

#ifndef SVT_SPLIT_TRACE
#define SVT_SPLIT_TRACE 0
#endif

namespace svt {

// the split region of the workgroup's LDS, byte offsets from StreamArgs::lds_rings (128-byte aligned)
constexpr uint32_t kSplitTileAt = 0;                                  // uint4[256]  {first record, records, sub2, flags} by sorted position
constexpr uint32_t kSplitUnitAt = kSplitTileAt + kBlock * 16u;        // uint32[256] unit index (kPadUnit: none)
constexpr uint32_t kSplitGateAt = kSplitUnitAt + kBlock * 4u;         // uint32[256] library windows: bit l = the small-deletion gate of the window's l-th library is closed
constexpr uint32_t kSplitTallyAt = kSplitGateAt + kBlock * 4u;        // double[5][256] tallies by sorted position
constexpr uint32_t kSplitRingAt = kSplitTallyAt + 5u * kBlock * 8u;   // per-wave rings: 4 K waves x (64 / K units x 128 bytes) = 32 KB; sort scratch
                                                                      // before the steps, the four result rings after them
constexpr uint32_t kSplitRegionBytes = kSplitRingAt + (uint32_t)(kBlock / kWave) * kRingBytes;
static_assert(kSplitRingAt % 128u == 0u, "rings are line-aligned");
constexpr uint32_t kSplitFmask = 7u, kSplitDel16 = 16u, kSplitSvtypeShift = 8u, kSplitUflagsShift = 16u;

typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4_s;

// one register of every lane from the lane `shift` places below it in its DPP row (shift > 0) / above it (shift < 0)
template <int SHIFT>
__device__ __forceinline__ uint32_t dpp_row_shift(const uint32_t v)
{
    // DPP control: row_shr:n = 0x110 + n (lane i reads lane i - n), row_shl:n = 0x100 + n (lane i reads lane i + n)
    constexpr int ctrl = SHIFT > 0 ? 0x110 + SHIFT : 0x100 - SHIFT;
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, 0xf, 0xf, false);
}
template <int SHIFT>
__device__ __forceinline__ double dpp_row_shift_f64(const double x)
{
    return __hiloint2double((int)dpp_row_shift<SHIFT>((uint32_t)__double2hiint(x)), (int)dpp_row_shift<SHIFT>((uint32_t)__double2loint(x)));
}

// the running sums a group hands from lane to lane
template <bool SSO>
struct SplitSums {
    double ref_seq, alt_seq, alt_clip, ref_span, alt_span;
    double l_ref_seq, l_alt_seq, l_alt_clip;   // sso: fragment-local sums (singlesample.py:246-276)
    template <int SHIFT>
    __device__ __forceinline__ SplitSums shifted() const
    {
        SplitSums t;
        t.ref_seq = dpp_row_shift_f64<SHIFT>(ref_seq);
        t.alt_seq = dpp_row_shift_f64<SHIFT>(alt_seq);
        t.alt_clip = dpp_row_shift_f64<SHIFT>(alt_clip);
        t.ref_span = dpp_row_shift_f64<SHIFT>(ref_span);
        t.alt_span = dpp_row_shift_f64<SHIFT>(alt_span);
        if (SSO) {
            t.l_ref_seq = dpp_row_shift_f64<SHIFT>(l_ref_seq);
            t.l_alt_seq = dpp_row_shift_f64<SHIFT>(l_alt_seq);
            t.l_alt_clip = dpp_row_shift_f64<SHIFT>(l_alt_clip);
        } else {
            t.l_ref_seq = t.l_alt_seq = t.l_alt_clip = 0.0;
        }
        return t;
    }
};

// what a record adds (record_single's look-ups and products, svt_stream_kernel.h)
struct SplitAddends {
    double rs_a, rs_b, p_seq, p_clip, alt_w, ref_w;
    bool cont;
};

template <bool SSO, int MODE, int K>
__global__ __launch_bounds__(kBlock * K, 4) void svt_split_kernel(const StreamArgs a)
{
    static_assert(MODE == kSingleLds || MODE == kMultiLds, "one library, or library windows (svt_unit.libs)");
    static_assert(K == 2 || K == 4, "two or four lanes per unit");
    constexpr uint32_t kWaves = 4u * K, kThreads = kWaves * kWave, kUnitsPerWave = kWave / K, kRecs = kBlockRecords / K;
    constexpr uint32_t kWaveRing = kUnitsPerWave * 128u;   // one 128-byte block per unit of the wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();   // tables at absolute LDS addresses
    const uint32_t tid = threadIdx.x, lane = tid % kWave;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));
    unsigned char* region = smem + a.lds_rings;
    uint4* s_tile = reinterpret_cast<uint4*>(region + kSplitTileAt);
    uint32_t* s_unit = reinterpret_cast<uint32_t*>(region + kSplitUnitAt);
    uint32_t* s_gate = reinterpret_cast<uint32_t*>(region + kSplitGateAt);
    double* s_tally = reinterpret_cast<double*>(region + kSplitTallyAt);
    unsigned char* rings = region + kSplitRingAt;

    // this workgroup's units: 256 consecutive ones, or (library windows) a chunk of at most 256 units of the permutation that groups
    // the units by the libraries of their sample
    const uint32_t wg_index = MODE == kMultiLds ? blockIdx.x + a.chunk_begin : blockIdx.x;
    uint32_t wg_base = a.unit_begin + blockIdx.x * a.units_per_wg, n_here;
    WgDesc wd{};
    if (MODE == kMultiLds) {
        const uint2 ch = a.chunks[wg_index];
        wg_base = ch.x;
        n_here = min(ch.y, (uint32_t)kBlock);
        wd = a.windows[wg_index];
    } else {
        n_here = min(a.units_per_wg, a.unit_end - wg_base);
    }
    const bool sorter = tid < (uint32_t)kBlock;    // the first four waves hold the workgroup's (up to) 256 units

    // ---- this thread's unit: record range and header (the loads overlap the table staging below)
    uint32_t beg = 0u, cnt = 0u, my_unit = kPadUnit;
    svt_unit U{};
    if (sorter && tid < n_here) {
        my_unit = MODE == kMultiLds && a.perm ? a.perm[wg_base + tid] : wg_base + tid;
        const uint64_t lo = a.rec_offset[my_unit], hi = a.rec_offset[my_unit + 1];
        beg = (uint32_t)lo;
        cnt = (uint32_t)(hi - lo);
        U = a.units[my_unit];
    }
    // ---- tables (the layout of svt_stream_kernel: kSPm ...)
    for (uint32_t i = tid; i < 256; i += kThreads) {
        const double p = a.pm[i];
        reinterpret_cast<double*>(smem + kSPm)[i] = p;
        reinterpret_cast<double*>(smem + kSPmHalf)[i] = p * 0.5;
    }
    if (tid < 32) {
        const PairWeights pw = a.wtab[tid];
        reinterpret_cast<double*>(smem + kSWtab)[tid] = pw.w_alt;
        reinterpret_cast<double*>(smem + kSWtab + kSWref)[tid] = pw.w_ref;
    }
    if (MODE == kSingleLds) {
        int16_t* s_thr = reinterpret_cast<int16_t*>(smem + kSBins);
        uint16_t* s_hst = reinterpret_cast<uint16_t*>(smem + kSBins) + a.total_bins;
        for (uint32_t i = tid; i < a.total_bins; i += kThreads) {
            const Bin bn = a.bins[i];
            s_thr[i] = (int16_t)bn.thr;
            s_hst[i] = (uint16_t)bn.hist;
        }
    } else {
        // the window's bins as thr[bin_cnt], hist[bin_cnt] and one WinLib per library of the window (svt_stream_kernel.h)
        int16_t* s_thr = reinterpret_cast<int16_t*>(smem + kSBins);
        uint16_t* s_hst = reinterpret_cast<uint16_t*>(smem + kSBins) + wd.bin_cnt;
        for (uint32_t i = tid; i < wd.bin_cnt; i += kThreads) {
            const Bin bn = a.bins[wd.bin_lo + i];
            s_thr[i] = (int16_t)bn.thr;
            s_hst[i] = (uint16_t)bn.hist;
        }
        if (tid < wd.lib_cnt) {
            const LibDesc L = a.libs[wd.lib_lo + tid];
            WinLib wl;
            wl.kmin = (uint32_t)L.key_min;
            wl.nb = L.n_bins;
            wl.thr_at = kSBins + (L.tab_off - wd.bin_lo) * 2u;
            wl.hist_at = kSBins + (wd.bin_cnt + L.tab_off - wd.bin_lo) * 2u;
            wl.sd2 = L.sd2;
            wl.pad = 0.0;
            reinterpret_cast<WinLib*>(smem + a.lds_winlibs)[tid] = wl;
        }
    }
    if (a.l10_where == kL10Shared) {
        double* s_l10 = reinterpret_cast<double*>(smem + a.lds_l10);
        for (uint32_t i = tid; i < a.n_l10; i += kThreads) s_l10[i] = a.l10[i];
    }
    for (uint32_t i = tid; i < 5u * kBlock; i += kThreads) s_tally[i] = 0.0;

    // ---- counting sort of the units by block count, longest first (wg_sort_into_tiles; the other waves only meet the barriers).
    // What moves is the tile entry: everything a record's look-ups need to know of its unit (record_single's StreamCtx).
    {
        uint32_t* s_hist = reinterpret_cast<uint32_t*>(rings);
        uint32_t* s_start = s_hist + (kMaxSortKey + 1);
        uint32_t* s_wsum = s_start + (kMaxSortKey + 1);
        const uint32_t nblk = cnt ? ((beg & 7u) + cnt + 7u) >> 3 : 0u;
        // (a unit without records still sorts in front of the padding threads: the workgroup's units fill its first positions)
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
            s_start[kMaxSortKey - tid] = before + incl - h;
        }
        __syncthreads();
        if (sorter) {
            const bool is_del = U.svtype == SVT_SVTYPE_DEL;
            const bool small_del = MODE == kSingleLds && is_del && ((double)U.pos_delta < a.lib0.sd2);   // classic.py:339,383
            const uint32_t flags = (small_del ? 0u : kSplitFmask) | (is_del ? kSplitDel16 : 0u) | ((uint32_t)U.svtype << kSplitSvtypeShift) |
                                   ((uint32_t)U.flags << kSplitUflagsShift);
            // one library: DEL ? var_length + key_min : never in range; windows: DEL ? var_length : never -- the library's key_min is added per record
            const uint32_t sub2 = is_del ? (uint32_t)U.var_length + (MODE == kSingleLds ? (uint32_t)a.lib0.key_min : 0u) : 0x80000000u;
            const uint32_t pos = s_start[key] + rank;
            s_tile[pos] = make_uint4(beg, cnt, sub2, flags);
            s_unit[pos] = my_unit;
            if (MODE == kMultiLds) {   // the small-deletion gate per library of the window, formed once per unit (the WinLibs were staged before the sort's barriers)
                uint32_t gated = 0u;
                if (is_del)
                    for (uint32_t l = 0; l < wd.lib_cnt; ++l)
                        gated |= ((double)U.pos_delta < reinterpret_cast<const WinLib*>(smem + a.lds_winlibs)[l].sd2 ? 1u : 0u) << l;
                s_gate[pos] = gated;
            }
        }
        __syncthreads();
    }

    // ---- this lane: lane h of the group of unit `pos`
    const uint32_t hh = lane % K, pos = wave * kUnitsPerWave + lane / K;
    const uint4 me = s_tile[pos];
    const uint32_t first_rec = me.x, n_rec = me.y;
    const uint32_t head = first_rec & 7u, last = head + n_rec;
    const uint32_t nblk = n_rec ? (last + 7u) >> 3 : 0u;
    uint32_t max_blk = nblk;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) max_blk = max(max_blk, (uint32_t)__shfl_xor((int)max_blk, d, kWave));
    max_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)max_blk);
    uint32_t min_blk = nblk;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) min_blk = min(min_blk, (uint32_t)__shfl_xor((int)min_blk, d, kWave));
    min_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)min_blk);
    StreamCtx sc;
    sc.fmask = me.w & kSplitFmask;
    sc.kmin = (uint32_t)a.lib0.key_min;
    sc.nb = a.lib0.n_bins;
    sc.sub2 = me.z;
    sc.hist_at = kSBins + a.total_bins * 2u;
    sc.wt0 = kSWtab + (me.w & kSplitDel16) * 8u;
    sc.wt1 = sc.wt0 + 8u * 8u;
    sc.wh0 = 0u;
    const uint32_t gated = MODE == kMultiLds ? s_gate[pos] : 0u;
    const bool is_del = (me.w & kSplitDel16) != 0u;
    const uint32_t lib_last = wd.lib_cnt - 1u, winlibs_at = a.lds_winlibs;
    const uint32_t neutral_w = MODE == kMultiLds ? wd.lib_lo << SVT_REC_LIB_SHIFT : 0u;
    const uint32_t lib_key = MODE == kMultiLds && wd.lib_cnt == 1u ? wd.lib_lo << SVT_REC_LIB_SHIFT : 0u;

    // ---- fetch side: lane (o, rr) of instruction i brings 16 bytes of the block of the wave's unit 8 i + o to ring + (8 i + o) * 128 + rr * 16;
    // the slot rr of unit u holds logical record rr ^ swz(u), swz(u) = (u >> 1) & 7 (svt_ring_engine.h)
    unsigned char* ring = rings + wave * kWaveRing;
    const uint32_t ring_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const uint32_t o = lane >> 3, rr = lane & 7u;
    constexpr int kFetches = (int)kUnitsPerWave / 8;
    uint32_t src_first[kFetches], src_end[kFetches], src_base[kFetches];
#pragma unroll
    for (int i = 0; i < kFetches; ++i) {
        const uint32_t u = 8u * (uint32_t)i + o;
        const uint4 e = s_tile[wave * kUnitsPerWave + u];
        src_first[i] = e.x;
        src_end[i] = e.x + e.y;
        src_base[i] = (e.x & ~7u) + (rr ^ ((u >> 1) & 7u));
    }
    const char* rec_bytes = reinterpret_cast<const char*>(a.records);
    auto fetch = [&](const uint32_t k) {
        if (k >= 1u && k + 1u < min_blk) {   // an interior block of every unit of the wave: whole lines, nothing to test
#pragma unroll
            for (int i = 0; i < kFetches; ++i) {
                const uint32_t rec = src_base[i] + k * kBlockRecords;
                __builtin_amdgcn_global_load_lds(rec_bytes + ((uint64_t)rec << 4), (lds_void_ptr)(ring + (uint32_t)i * 1024u), 16, 0, SVT_STREAM_AUX);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < kFetches; ++i) {
            const uint32_t rec = src_base[i] + k * kBlockRecords;
            if (rec >= src_first[i] && rec < src_end[i])
                __builtin_amdgcn_global_load_lds(rec_bytes + ((uint64_t)rec << 4), (lds_void_ptr)(ring + (uint32_t)i * 1024u), 16, 0, SVT_STREAM_AUX);
        }
    };
    // consumer side: logical record h * kRecs + j of this lane's unit sits in slot (h * kRecs + j) ^ swz(unit in wave)
    const uint32_t uw = lane / K;
    uint32_t rd[kRecs];
#pragma unroll
    for (uint32_t j = 0; j < kRecs; ++j) rd[j] = ring_addr + uw * 128u + (((hh * kRecs + j) ^ ((uw >> 1) & 7u)) << 4);

    RecordCheck<MODE> check;
    SplitSums<SSO> run = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // the group's running sums; between blocks: in lane 0
    if (max_blk) fetch(0u);
    for (uint32_t k = 0; k < max_blk; ++k) {
        u32x4 w[kRecs];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (uint32_t j = 0; j < kRecs; ++j) w[j] = *reinterpret_cast<lds_cu32x4_s*>((size_t)rd[j]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (k + 1 < max_blk) fetch(k + 1);
        // ---- what this lane's records add; a slot outside the unit is the neutral record (MAPQ 0 everywhere adds +0.0).
        // Two records at a time, in three steps with one LDS round trip each -- every look-up that depends on the record alone, the
        // decision-table look-up that depends on p_concordant, the products -- instead of six round trips per record: a wave
        // that has its SIMD nearly to itself waits for every one of them.
        SplitAddends x[kRecs];
        bool any_cont = false;
#pragma unroll
        for (uint32_t j0 = 0; j0 < kRecs; j0 += 2u) {
            u32x4 wj[2];
            double pm_a[2], pm_b[2], s0[2], s1[2], c0[2], c1[2];
            int32_t thr1[2];
            uint32_t h2[2], f3[2];
#pragma unroll
            for (uint32_t q = 0; q < 2u; ++q) {
                const uint32_t j = j0 + q;
                const bool mine = k * kBlockRecords + hh * kRecs + j - head < n_rec;   // head <= index < last, unsigned
                wj[q].x = mine ? w[j].x : 0u;
                wj[q].y = mine ? w[j].y : 0u;
                wj[q].z = mine ? w[j].z : 0u;
                wj[q].w = mine ? w[j].w : neutral_w;
                check.see(wj[q], lib_key);
                if (MODE == kMultiLds) {
                    // the record's library picks one of the window's descriptors (record_window, svt_stream_kernel.h)
                    const uint32_t rel = SVT_REC_LIB(wj[q].w) - wd.lib_lo;
                    check.window_lib(rel);
                    const uint32_t l = min(rel, lib_last);
                    const u32x4 d = *reinterpret_cast<lds_cu32x4_s*>((size_t)(winlibs_at + l * (uint32_t)sizeof(WinLib)));   // kmin, nb, thr_at, hist_at
                    f3[q] = ((gated >> l) & 1u) ? 0u : (wj[q].w & 7u);          // classic.py:339,383
                    const uint32_t sub2 = is_del ? sc.sub2 + d.x : 0x80000000u;
                    const uint32_t i1 = min(wj[q].x - d.x, d.y), i2 = min(wj[q].x - sub2, d.y);
                    thr1[q] = lds_i16(d.z + (i1 << 1));
                    h2[q] = lds_u16(d.w + (i2 << 1));
                } else {
                    f3[q] = wj[q].w & sc.fmask;
                    const uint32_t i1 = min(wj[q].x - sc.kmin, sc.nb), i2 = min(wj[q].x - sc.sub2, sc.nb);
                    thr1[q] = lds_i16(kSBins + (i1 << 1));
                    h2[q] = lds_u16(sc.hist_at + (i2 << 1));
                }
                pm_a[q] = lds_f64(kSPm + byte0_x8(wj[q].y));
                pm_b[q] = lds_f64(kSPm + byte1_x8(wj[q].y));
                x[j].rs_a = lds_f64(kSPm + byte2_x8(wj[q].y));
                x[j].rs_b = lds_f64(kSPm + byte3_x8(wj[q].y));
                s0[q] = lds_f64(kSPmHalf + byte0_x8(wj[q].z));
                s1[q] = lds_f64(kSPmHalf + byte1_x8(wj[q].z));
                c0[q] = lds_f64(kSPmHalf + byte2_x8(wj[q].z));
                c1[q] = lds_f64(kSPmHalf + byte3_x8(wj[q].z));
            }
            __builtin_amdgcn_sched_barrier(0);
            double w_alt[2], w_ref[2];
#pragma unroll
            for (uint32_t q = 0; q < 2u; ++q) {
                const bool p_conc = (int32_t)h2[q] <= thr1[q];
                const uint32_t wa = (p_conc ? sc.wt1 : sc.wt0) | (f3[q] << 3);   // &w_alt[f3 | p_conc << 3 | del16]
                w_alt[q] = lds_f64(wa);
                w_ref[q] = lds_f64(wa + kSWref);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (uint32_t q = 0; q < 2u; ++q) {
                const uint32_t j = j0 + q;
                const double pp = pm_a[q] * pm_b[q];
                x[j].p_seq = s0[q] + s1[q];
                x[j].p_clip = c0[q] + c1[q];
                x[j].alt_w = pp * w_alt[q];
                x[j].ref_w = pp * w_ref[q];
                x[j].cont = SSO && (wj[q].w & SVT_REC_CONTINUATION) != 0u;
                any_cont = any_cont || x[j].cont;
            }
        }
        const bool has_cont = SSO && __any(any_cont);
        // ---- the sums, in record order: lane 0 of the group, then lane 1 with what lane 0 hands over, ...
#pragma unroll
        for (uint32_t p = 0; p < (uint32_t)K; ++p) {
            if (hh == p) {
#pragma unroll
                for (uint32_t j = 0; j < kRecs; ++j) {
                    if (!SSO) {                     // classic.py:306-328
                        run.ref_seq = (run.ref_seq + x[j].rs_a) + x[j].rs_b;
                        run.alt_seq += x[j].p_seq;
                        run.alt_clip += x[j].p_clip;
                    } else if (!has_cont) {         // no record of the block continues a fragment: record_weights<SSO, ., false>
                        run.ref_seq += run.l_ref_seq;
                        run.alt_seq += run.l_alt_seq;
                        run.alt_clip += run.l_alt_clip;
                        run.l_ref_seq = x[j].rs_a + x[j].rs_b;
                        run.l_alt_seq = x[j].p_seq;
                        run.l_alt_clip = x[j].p_clip;
                    } else {                        // singlesample.py:246-276,367-372: the general form of record_weights<SSO>
                        const bool cont = x[j].cont;
                        run.ref_seq += cont ? 0.0 : run.l_ref_seq;
                        run.alt_seq += cont ? 0.0 : run.l_alt_seq;
                        run.alt_clip += cont ? 0.0 : run.l_alt_clip;
                        run.l_ref_seq = ((cont ? run.l_ref_seq : 0.0) + x[j].rs_a) + x[j].rs_b;
                        run.l_alt_seq = (cont ? run.l_alt_seq : 0.0) + x[j].p_seq;
                        run.l_alt_clip = (cont ? run.l_alt_clip : 0.0) + x[j].p_clip;
                    }
                    run.alt_span += x[j].alt_w;
                    run.ref_span += x[j].ref_w;
                }
            }
            // hand over: lane p + 1 takes lane p's sums; after the group's last lane they go back to lane 0.  (Every lane takes its
            // neighbour's registers: only the lane whose turn is next will read them before it is handed something itself.)
            run = p + 1u < (uint32_t)K ? run.template shifted<1>() : run.template shifted<-(K - 1)>();
        }
    }
    // ---- the unit's tallies (in lane 0 of its group) wait in LDS for the epilogues
    if (hh == 0u) {
        if (SSO) {  // flush the last fragment (singlesample.py:370-372)
            run.ref_seq += run.l_ref_seq;
            run.alt_seq += run.l_alt_seq;
            run.alt_clip += run.l_alt_clip;
        }
        s_tally[pos] = run.ref_seq;
        s_tally[kBlock + pos] = run.alt_seq;
        s_tally[2 * kBlock + pos] = run.alt_clip;
        s_tally[3 * kBlock + pos] = run.ref_span;
        s_tally[4 * kBlock + pos] = run.alt_span;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // every unit's tallies are in LDS; the rings are free for the result records

    // ---- epilogues: wave w takes sorted positions 64 w ... (classic.py:425-513 in unit_epilogue)
    if (wave < (uint32_t)kWavesPerBlock) {
        const uint32_t p = wave * kWave + lane;
        Acc acc = {s_tally[p], s_tally[kBlock + p], s_tally[2 * kBlock + p], s_tally[3 * kBlock + p], s_tally[4 * kBlock + p], 0.0, 0.0, 0.0};
        const uint32_t unit = s_unit[p], flags = s_tile[p].w;
        const double* lds_l10 = reinterpret_cast<const double*>(smem + a.lds_l10);
        uint4 piece[8];
        unit_epilogue(acc, (flags >> kSplitSvtypeShift) & 0xffu, flags >> kSplitUflagsShift, a.c, lds_l10, a.l10, a.l10_where == kL10Shared ? a.l10_lds_entries : 0u, piece);
        uint32_t unit_out = unit;
        if (a.out_samples > 1u && unit != kPadUnit) {
            const uint32_t sample = unit / a.out_sites;
            unit_out = (unit - sample * a.out_sites) * a.out_samples + sample;
        }
        uint32_t tile_slot = 0xFFFFFFFFu;
        if (a.result96) {
            piece[5] = make_uint4(piece[5].x, piece[7].y, unit_out, 0u);
            tile_slot = a.slot_begin + (wg_index * (uint32_t)kWavesPerBlock + wave) * kWave;
        }
        store_result_records_through_ring(rings + wave * kRingBytes, piece, unit_out, lane, reinterpret_cast<unsigned char*>(a.out), a.result96 ? 6u : 8u, tile_slot);
    }
    const uint32_t bad = check.bits(MODE == kMultiLds ? wd.lib_cnt : a.n_libs);
    if (bad) atomicOr(a.err, bad);
}

}  // namespace svt

#endif  // SVT_SPLIT_KERNEL_H
