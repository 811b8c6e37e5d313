// svt_stream_kernel.h -- the genotype pass over the canonical CSR records as they are (no re-tiling)
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
//
// One kernel launch takes the section-8(d) canonical input -- rec_offset[], svt_unit[], svt_record[] in the
// order the caller packed them -- to the 128-byte result records: tally -> zeroing rules -> QR/QA ->
// bayes_gt -> GT/GQ/SQ (svtyper/classic.py:296-513, singlesample.py:246-473).  Every record is read from
// HBM exactly once, nothing is re-encoded, sorted or tiled on the host or in a separate kernel.
//
// The five tallies are sequential binary64 sums in record order, so a unit still belongs to ONE lane.  What makes
// that coalesced is the per-wave LDS ring of svt_ring_engine.h (workgroup length sort, LDS-DMA fetch of one 128-byte
// line per unit and step, XOR-swizzled conflict-free reads, results written as full lines through the same ring).
// This file adds the record arithmetic on top of it:
//
//   * record_single  one library: every table at a fixed LDS address, every field of a record turned into a
//                    ds_read address by ONE instruction (SDWA byte select + immediate table base);
//   * record_window  several libraries with per-sample library windows (svt_unit.libs): the units are grouped by
//                    window on the host (a permutation of unit indices, the records stay where they are) and a
//                    workgroup stages only its window's histograms;
//   * the general mode (any geometry, no hints): weight_evidence / pair_evidence with tables through L2;
//   * slots of a block that lie outside the unit (the neighbours' records in its first and last line, everything
//     past the end of a shorter unit) are not fetched; the consumer reads them as records with MAPQ 0 everywhere:
//     prob_mapq(0) == +0.0 exactly, and x + 0.0 == x for these non-negative sums (the argument of
//     include/svtyper_hip.h for gated-off reads).
//
// The record contract of include/svtyper_hip.h is checked on the fly: violations are OR-ed into *err, which the
// host reads after the pass.
#ifndef SVT_STREAM_KERNEL_H
#define SVT_STREAM_KERNEL_H

// Tunables (each the winner of in-process A/B runs; the measured alternatives, including the ones whose switches were removed
// from this file in round 6 -- one-trip table reads, an edge / interior consumer pair, direct stores, per-step edge policies --
// are in profiles/HISTORY_design_r03_r05.md).  tools/stream_variants.sh builds variants with -D.
#ifndef SVT_STREAM_AUX
#define SVT_STREAM_AUX 2      // cache policy bits of the record fetches (2 = nt: every line is used once)
#endif
#ifndef SVT_STREAM_EDGE_AUX
#define SVT_STREAM_EDGE_AUX 0 // ... of the blocks that hold a unit's first / last line (0 = default: the neighbour's request may hit L2)
#endif
#ifndef SVT_STREAM_SPLIT
#define SVT_STREAM_SPLIT 4    // scheduling barrier in front of this record of a block (8 = none)
#endif
#ifndef SVT_STREAM_WAVES
#define SVT_STREAM_WAVES 3    // one library: waves per SIMD the register allocation must allow (the kernel needs 118 VGPRs: four fit)
#endif
#ifndef SVT_WINDOW_WAVES
#define SVT_WINDOW_WAVES 3    // library windows: the same (161 VGPRs with both record consumers: three workgroups per CU)
#endif

#include <type_traits>

#include "svt_ring_engine.h"

namespace svt {

// LDS layout of the streaming kernel, absolute byte addresses (the kernel has no static LDS, so the dynamic
// segment starts at 0 -- checked at run time): the one-library record consumer turns every field of a record
// into an LDS address with one instruction and the table base as the ds_read's immediate offset.
constexpr uint32_t kSPm = 0;                          // double[256]  prob_mapq(q)                      (utils.py:74-75)
constexpr uint32_t kSPmHalf = kSPm + 256 * 8;         // double[256]  prob_mapq(q) / 2 (exact: a power-of-two scaling)
constexpr uint32_t kSWtab = kSPmHalf + 256 * 8;       // kSingleLds: double w_alt[32], w_ref[32] (columns); kGeneral: PairWeights[32]
constexpr uint32_t kSWref = 32 * 8;                   // byte distance w_alt[i] -> w_ref[i]
constexpr uint32_t kSWhi = kSWtab + 2 * 32 * 8;       // kSingleLds / kMultiLds: uint32 high words of w_alt[32], w_ref[32] (their low words are 0)
constexpr uint32_t kSWhiRef = 32 * 4;                 // byte distance w_alt_hi[i] -> w_ref_hi[i]
constexpr uint32_t kSBins = kSWhi + 2 * 32 * 4;       // kSingleLds: int16 thr[total_bins], uint16 hist[total_bins] (ranks, svt_host_tables.h); kGeneral: LibDesc[n_libs]

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
    uint32_t l10_where;          // kL10Shared / kL10Ring / kL10Global
    uint32_t lds_l10;            // kL10Shared: byte offset of the workgroup's copy of the log10 table
    uint32_t l10_lds_entries;    // entries of the table the epilogue finds in LDS (kL10Shared: all; kL10Ring: what one ring stage holds --
                                 // a unit whose read count reaches beyond takes the table through L2; kL10Global: 0)
    uint64_t n_units;
    svt_result* out;
    uint32_t* err;
    // kMultiLds: units grouped by the library window of their sample (svt_unit.libs)
    const uint32_t* perm;        // unit indices, grouped by window, original order inside a group; nullptr = the identity
    const uint2* chunks;         // one per workgroup: {first position in perm, units (<= 256 * R)} -- never crosses a group
    const WgDesc* windows;       // one per workgroup: the libraries / bins it stages
    uint32_t lds_winlibs;        // byte offset of the WinLib descriptors (after the bins)
    uint32_t unit_begin;         // this launch covers units [unit_begin, unit_end) (the pipelined one-shot launches
    uint32_t unit_end;           // one range per uploaded piece; a pass over a resident batch: [0, n_units))
    uint32_t units_per_wg;       // (not the window mode) consecutive units of one workgroup, <= 256 * R: the host cuts a launch into
                                 // EQUAL workgroups that fill whole rounds of the chip's resident workgroups (svtyper_hip.hip: wg_plan)
    uint32_t chunk_begin;        // (library windows) this launch covers the chunks from this one on
    uint32_t result96;           // SVT_FLAG_RESULT96: `out` holds 96-byte records (svt_result96) in the workgroups' own order, tagged with their unit
    uint32_t slot_begin;         // ... the first of them this launch writes (workgroup w of the launch: slot_begin + w * 256 * R)
    uint32_t out_samples;        // svt_batch_result_order: > 1 = the units are sample-major (unit = sample * out_sites + site) and the
    uint32_t out_sites;          // result record of a unit goes to index site * out_samples + sample (site-major); 0 = unit order
    LibDesc lib0;
    GtConsts c;
};

// kMultiLds: one library of the workgroup's window, 32 bytes in LDS
struct WinLib {
    uint32_t kmin;      // (uint32) key_min
    uint32_t nb;        // n_bins == index of the library's sentinel bin
    uint32_t thr_at;    // LDS byte address of this library's thr[0]
    uint32_t hist_at;   // LDS byte address of this library's hist[0]
    double sd2;         // 2 * sd: the small-deletion gate (classic.py:339,383)
    double pad;
};
static_assert(sizeof(WinLib) == 32, "WinLib is read as two 16-byte halves");

// The record contract of include/svtyper_hip.h, accumulated over the records of a lane's units at 3-4
// instructions per record.
template <int MODE>
struct RecordCheck {
    uint32_t flags_or = 0, span_or = 0, lone = 0, lib_max = 0;
    // kMultiLds: `lib_key` = the library byte every record must carry when the window holds ONE library (else 0)
    __device__ __forceinline__ void see(const u32x4 w, const uint32_t lib_key = 0u)
    {
        if (MODE == kMultiLds) flags_or |= w.w ^ lib_key;   // a one-library window: the library byte must cancel
        else flags_or |= w.w;                                // undefined bits; with one library also the library byte
        span_or |= w.x;                                      // sign bit: a negative ospan_len
        lone = max(lone, (w.w & 0x17u) ^ 0x10u);             // > 0x10: straddle bits without HAS_PAIR
        if (MODE == kGeneral) lib_max = max(lib_max, w.w & 0xffff00u);
    }
    // kMultiLds, windows of several libraries: the consumer has `library - first library of the window` at hand
    // (unsigned: below the window = huge).  One more loop-carried value in see() costs the kernel its third wave.
    __device__ __forceinline__ void window_lib(const uint32_t d) { lib_max = max(lib_max, d); }
    // kMultiLds: `limit` = libraries in the unit's window
    __device__ __forceinline__ uint32_t bits(const uint32_t limit) const
    {
        const uint32_t n_libs = limit;
        const bool bad_lib = MODE == kSingleLds ? (flags_or & 0xffff00u) != 0u
                             : MODE == kMultiLds ? (limit == 1u ? (flags_or & 0xffff00u) != 0u : lib_max >= limit)
                                                 : (lib_max >> SVT_REC_LIB_SHIFT) >= n_libs;
        return (lone > 0x10u ? kErrStraddleNoPair : 0u) | (bad_lib ? kErrLibIndex : 0u) |
               ((flags_or & ~SVT_REC_FLAG_MASK) ? kErrReservedBits : 0u) | ((int32_t)span_or < 0 ? kErrNegativeSpan : 0u);
    }
};

// per-lane constants of the unit for the one-library record consumer
struct StreamCtx {
    uint32_t fmask;    // straddle-bit mask with the small-DEL gate applied (classic.py:339,383)
    uint32_t kmin;     // (uint32) key_min
    uint32_t nb;       // n_bins == index of the sentinel bin
    uint32_t sub2;     // DEL ? var_length + key_min : 0x80000000 (never in range)
    uint32_t hist_at;  // LDS address of hist[0]
    uint32_t wt0, wt1; // LDS address of w_alt[del16] / w_alt[del16 + 8] (p_concordant = 0 / 1)
    uint32_t wh0;      // LDS address of w_alt_hi[del16]
};

// One canonical record, one library, tables at fixed LDS addresses: the arithmetic of weight_evidence +
// pair_evidence (svt_unit_math.h; classic.py:306-408) with every table index formed by one
// instruction.  (pm(l) * L + pm(r) * R) / 2.0 (classic.py:324) is taken as pm(l)/2 + pm(r)/2 from a second
// table: halving a binary64 in [0.2, 1] is exact, so the sum rounds identically.  EDGE: the record may belong
// to a neighbouring unit -- its weight bytes are then read as MAPQ 0, which adds +0.0 to every sum.
// the split-read / reference-read part of a record (classic.py:306-328); returns pm(mapq_a) * pm(mapq_b)
// CONT = false (sso only): no lane's record of this block continues a fragment, so every record starts one: the
// previous fragment's sums go to the site totals and the new ones start from 0.0 + x == x (x >= +0.0) -- the same
// values as the general form below without its selects.
template <bool SSO, bool EDGE, bool CONT = true>
__device__ __forceinline__ double record_weights(const u32x4 w, const bool mine, Acc& a)
{
    const uint32_t wy = EDGE ? (mine ? w.y : 0u) : w.y;   // mapq_a | mapq_b << 8 | rs_a << 16 | rs_b << 24
    const uint32_t wz = EDGE ? (mine ? w.z : 0u) : w.z;   // seq_l | seq_r << 8 | clip_l << 16 | clip_r << 24
    const double pm_a = lds_f64(kSPm + byte0_x8(wy)), pm_b = lds_f64(kSPm + byte1_x8(wy));
    const double rs_a = lds_f64(kSPm + byte2_x8(wy)), rs_b = lds_f64(kSPm + byte3_x8(wy));
    const double p_seq = lds_f64(kSPmHalf + byte0_x8(wz)) + lds_f64(kSPmHalf + byte1_x8(wz));
    const double p_clip = lds_f64(kSPmHalf + byte2_x8(wz)) + lds_f64(kSPmHalf + byte3_x8(wz));
    if (SSO && !CONT) {
        a.ref_seq += a.l_ref_seq;
        a.alt_seq += a.l_alt_seq;
        a.alt_clip += a.l_alt_clip;
        a.l_ref_seq = rs_a + rs_b;
        a.l_alt_seq = p_seq;
        a.l_alt_clip = p_clip;
    } else if (SSO) {   // singlesample.py:246-276,367-372: per-fragment sums, added to the site totals when the next fragment starts
        const bool cont = (w.w & SVT_REC_CONTINUATION) != 0u;
        a.ref_seq += cont ? 0.0 : a.l_ref_seq;
        a.alt_seq += cont ? 0.0 : a.l_alt_seq;
        a.alt_clip += cont ? 0.0 : a.l_alt_clip;
        a.l_ref_seq = ((cont ? a.l_ref_seq : 0.0) + rs_a) + rs_b;
        a.l_alt_seq = (cont ? a.l_alt_seq : 0.0) + p_seq;
        a.l_alt_clip = (cont ? a.l_alt_clip : 0.0) + p_clip;
    } else {
        a.ref_seq = (a.ref_seq + rs_a) + rs_b;
        a.alt_seq += p_seq;
        a.alt_clip += p_clip;
    }
    return pm_a * pm_b;
}

template <bool SSO, bool EDGE, bool CONT>
__device__ __forceinline__ void record_single(const u32x4 w, const bool mine, const StreamCtx& c, Acc& a)
{
    const double pp = record_weights<SSO, EDGE, CONT>(w, mine, a);
    // p_concordant as the integer test hist[o - v] <= thr[o] (svt_host_tables.h), out-of-range -> sentinel bin
    const uint32_t i1 = min(w.x - c.kmin, c.nb), i2 = min(w.x - c.sub2, c.nb);
    const int32_t thr1 = lds_i16(kSBins + (i1 << 1));
    const uint32_t h2 = lds_u16(c.hist_at + (i2 << 1));
    const bool p_conc = (int32_t)h2 <= thr1;
    const uint32_t wa = (p_conc ? c.wt1 : c.wt0) | ((w.w & c.fmask) << 3);   // &w_alt[f3 | p_conc << 3 | del16]
    a.alt_span += pp * lds_f64(wa);
    a.ref_span += pp * lds_f64(wa + kSWref);
}

// per-lane constants of the unit for the library-window consumer
struct WindowCtx {
    uint32_t lib_lo;       // first library of the workgroup's window
    uint32_t lib_last;     // libraries in the window - 1
    uint32_t winlibs_at;   // LDS byte address of the window's WinLib descriptors
    uint32_t vl_or_never;  // DEL ? var_length : 0x80000000 -- key_min is added per library (never in range for the latter)
    uint32_t wt0, wt1;
    uint32_t wh0;          // LDS address of w_alt_hi[del16]
    uint32_t gated;        // bit l CLEAR: the small-deletion gate of classic.py:339,383 is closed for the window's l-th library (DEL and
                           // pos_delta < 2 sd of that library) -- per unit and library, so it is formed once per unit, not per record
    bool is_del;
};

typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;

// The same record with several libraries: the record's library picks one of the window's descriptors (a
// record that names a library outside its unit's window is reported by RecordCheck and reads the nearest one).
template <bool SSO, bool EDGE, bool CONT, class CHECK>
__device__ __forceinline__ void record_window(const u32x4 w, const bool mine, const WindowCtx& c, Acc& a, CHECK& check)
{
    const double pp = record_weights<SSO, EDGE, CONT>(w, mine, a);
    check.window_lib(EDGE && !mine ? 0u : SVT_REC_LIB(w.w) - c.lib_lo);
    const uint32_t la = c.winlibs_at + min(SVT_REC_LIB(w.w) - c.lib_lo, c.lib_last) * (uint32_t)sizeof(WinLib);
    const u32x4 d = *reinterpret_cast<lds_cu32x4*>((size_t)la);          // kmin, nb, thr_at, hist_at
    // the small-deletion gate (classic.py:339,383) as an all-ones / all-zeros word from one signed bit-field extract of the
    // unit's `open` bits; no select on is_del: for another svtype vl_or_never is 0x80000000 and o - (0x80000000 + key_min)
    // lies at or above 0x7fff0000 for every span the contract allows, i.e. in the sentinel bin as before
    const uint32_t open = (uint32_t)__builtin_amdgcn_sbfe((int32_t)c.gated, min(SVT_REC_LIB(w.w) - c.lib_lo, c.lib_last), 1u);
    const uint32_t f3 = w.w & 7u & open;
    const uint32_t sub2 = c.vl_or_never + d.x;
    const uint32_t i1 = min(w.x - d.x, d.y), i2 = min(w.x - sub2, d.y);
    const int32_t thr1 = lds_i16(d.z + (i1 << 1));
    const uint32_t h2 = lds_u16(d.w + (i2 << 1));
    const bool p_conc = (int32_t)h2 <= thr1;
    const uint32_t wa = (p_conc ? c.wt1 : c.wt0) | (f3 << 3);
    a.alt_span += pp * lds_f64(wa);
    a.ref_span += pp * lds_f64(wa + kSWref);
}

// WK (library windows only): 0 = a launch over windows of any size -- both record consumers in one kernel, 161 VGPRs, three
// workgroups per CU; 1 = every window of the launch holds ONE library (record_single only), 2 = every window holds several
// (record_window only): one consumer each, 126 VGPRs, four workgroups per CU.  The host launches the chunks of a batch as
// two ranges (chunks come ordered by the size of their window), StreamArgs::chunk_begin = the range's first chunk.
template <bool SSO, int MODE, int R, int WK = 0>
__global__ __launch_bounds__(kBlock, MODE == kSingleLds ? SVT_STREAM_WAVES : MODE == kMultiLds ? (WK ? 4 : SVT_WINDOW_WAVES) : 2) void svt_stream_kernel(const StreamArgs a)
{
    static_assert(WK == 0 || (MODE == kMultiLds && !SSO), "window kinds: classic library-window kernels only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // (lds_rings is 128-byte aligned)
    constexpr uint32_t kUnitsPerWg = kBlock * R;
    double* s_pm = reinterpret_cast<double*>(smem + kSPm);
    PairWeights* s_wtab = reinterpret_cast<PairWeights*>(smem + kSWtab);   // kGeneral
    LibDesc* s_lib = reinterpret_cast<LibDesc*>(smem + kSBins);            // kGeneral
    // the one-library consumer addresses the tables by absolute LDS byte offsets
    if (MODE != kGeneral && (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();
    unsigned char* rings = smem + a.lds_rings;
    const uint32_t tid = threadIdx.x, lane = tid % kWave;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / kWave));   // wave-uniform: ring addresses stay in SGPRs
    // this workgroup's units: 256 * R consecutive ones, or (library windows) a chunk of the permutation that groups
    // the units by the libraries of their sample
    uint32_t wg_base = a.unit_begin + blockIdx.x * a.units_per_wg, n_here;
    WgDesc wd{};
    const uint32_t wg_index = MODE == kMultiLds ? blockIdx.x + a.chunk_begin : blockIdx.x;   // (the launch's chunk range / workgroup)
    if (MODE == kMultiLds) {
        const uint2 ch = a.chunks[wg_index];
        wg_base = ch.x;
        n_here = ch.y;
        wd = a.windows[wg_index];
    } else {
        n_here = min(a.units_per_wg, a.unit_end - wg_base);
    }
    // (library windows: a.perm == nullptr when the units already come grouped by window)
    auto unit_at = [&](const uint32_t local) -> uint32_t { return MODE == kMultiLds && a.perm ? a.perm[wg_base + local] : wg_base + local; };

    // ---- this thread's R units: record range and sort key (the loads overlap the table staging below)
    uint32_t beg[R], cnt[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t local = (uint32_t)j * kBlock + tid;
        beg[j] = 0u;
        cnt[j] = 0u;
        if (local < n_here) {
            const uint32_t u = unit_at(local);
            const uint64_t lo = a.rec_offset[u], hi = a.rec_offset[u + 1];
            beg[j] = (uint32_t)lo;
            cnt[j] = (uint32_t)(hi - lo);
        }
    }

    // ---- stage the tables in LDS
    for (uint32_t i = tid; i < 256; i += kBlock) {
        const double p = a.pm[i];
        s_pm[i] = p;
        reinterpret_cast<double*>(smem + kSPmHalf)[i] = p * 0.5;
    }
    if (tid < 32) {
        const PairWeights pw = a.wtab[tid];
        if (MODE != kGeneral) {
            reinterpret_cast<double*>(smem + kSWtab)[tid] = pw.w_alt;
            reinterpret_cast<double*>(smem + kSWtab + kSWref)[tid] = pw.w_ref;
            reinterpret_cast<uint32_t*>(smem + kSWhi)[tid] = (uint32_t)__double2hiint(pw.w_alt);
            reinterpret_cast<uint32_t*>(smem + kSWhi + kSWhiRef)[tid] = (uint32_t)__double2hiint(pw.w_ref);
        } else {
            s_wtab[tid] = pw;
        }
    }
    if (MODE == kSingleLds) {
        // thr[] and hist[] as two 2-byte arrays (svt_host_tables.h replaced the counts by their ranks, which is
        // all `hist[o - v] <= thr[o]` needs): the random look-ups of a wave spread over every LDS bank
        int16_t* s_thr = reinterpret_cast<int16_t*>(smem + kSBins);
        uint16_t* s_hst = reinterpret_cast<uint16_t*>(smem + kSBins) + a.total_bins;
        for (uint32_t i = tid; i < a.total_bins; i += kBlock) {
            const Bin bn = a.bins[i];
            s_thr[i] = (int16_t)bn.thr;
            s_hst[i] = (uint16_t)bn.hist;
        }
    } else if (MODE == kMultiLds) {
        // the window's bins as thr[bin_cnt], hist[bin_cnt] and one WinLib per library of the window
        int16_t* s_thr = reinterpret_cast<int16_t*>(smem + kSBins);
        uint16_t* s_hst = reinterpret_cast<uint16_t*>(smem + kSBins) + wd.bin_cnt;
        for (uint32_t i = tid; i < wd.bin_cnt; i += kBlock) {
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
    } else {
        for (uint32_t i = tid; i < a.n_libs * (uint32_t)(sizeof(LibDesc) / 8); i += kBlock)
            reinterpret_cast<uint64_t*>(s_lib)[i] = reinterpret_cast<const uint64_t*>(a.libs)[i];
    }
    if (a.l10_where == kL10Shared) {
        double* s_l10 = reinterpret_cast<double*>(smem + a.lds_l10);
        for (uint32_t i = tid; i < a.n_l10; i += kBlock) s_l10[i] = a.l10[i];
    }
    // ---- counting sort of the workgroup's units by block count, longest first; R tiles per wave
    uint4 info[R];
    wg_sort_into_tiles<R>(rings, beg, cnt, n_here, tid, lane, wave, info);

    Tables t;   // kGeneral: tables through ordinary pointers, bins through L2
    t.pm = s_pm;
    t.wtab = s_wtab;
    t.libs = s_lib;
    t.bins = a.bins;

    unsigned char* ring = rings + wave * kStreamRingBytes;
    const uint32_t ring_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    // LDS-DMA: lane (o, rr) of instruction i fetches 16 bytes of the block of unit 8 i + o; they land at
    // ring + (8 i + o) * 128 + rr * 16.  The slot rr of unit u holds logical record rr ^ swz(u), swz(u) = (u >> 1) & 7
    // = (o >> 1) | (i & 1) << 2.
    const uint32_t o = lane >> 3, rr = lane & 7u;
    const uint32_t col_even = (rr ^ (o >> 1)) << 4, col_odd = col_even ^ 64u;
    // consumer: logical record j of this lane's block
    const uint32_t sw = (lane >> 1) & 7u;
    const uint32_t lane_block = ring_addr + lane * 128u, sw16 = sw << 4;
    const char* rec_bytes = reinterpret_cast<const char*>(a.records);

    RecordCheck<MODE> check;
    const uint32_t lib_key = MODE == kMultiLds && wd.lib_cnt == 1u ? wd.lib_lo << SVT_REC_LIB_SHIFT : 0u;

#pragma unroll      // (straight-line code: a loop lets LICM hoist the epilogue's ~40 constants into registers that then spill)
    for (int r = 0; r < R; ++r) {
        const uint32_t first_rec = info[r].x, n_rec = info[r].y;
        const uint32_t unit = info[r].z == kPadUnit ? kPadUnit : unit_at(info[r].z);
        svt_unit U{};
        if (unit != kPadUnit) U = a.units[unit];
        const uint32_t head = first_rec & 7u, last = head + n_rec;
        const uint32_t nblk = n_rec ? (last + 7u) >> 3 : 0u;
        uint32_t max_blk, min_blk;
        tile_block_range(nblk, max_blk, min_blk);
        // fetch side: lane (o, rr) of instruction i serves unit 8 i + o -- it needs that unit's record range
        uint32_t src_first[8], src_end[8], src_base[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            src_first[i] = (uint32_t)__shfl((int)first_rec, 8 * i + (int)o, kWave);
            src_end[i] = src_first[i] + (uint32_t)__shfl((int)n_rec, 8 * i + (int)o, kWave);
            src_base[i] = (src_first[i] & ~7u) + (((i & 1) ? col_odd : col_even) >> 4);
        }

        // interior blocks are read once and never again (non-temporal); a unit's first and last line are shared
        // with its neighbours in the CSR, which another wave fetches at another time: they can be given another
        // policy (SVT_STREAM_EDGE_AUX) so that the second request may be served by L2 / Infinity Cache
        // (measured: the second policy pays with two tiles per wave -- 512 neighbouring units per workgroup --, and
        // costs 2 % with one)
        constexpr int kEdgeAux = R >= 2 ? SVT_STREAM_EDGE_AUX : SVT_STREAM_AUX;
        auto fetch_first = [&]() {
            fetch_block<kEdgeAux, true>(0, src_base, src_first, src_end, rec_bytes, ring);
        };
        // block k (k >= 1) -> stage `st` of the ring; true when the group was exactly eight instructions (read_block's PENDING)
        auto fetch = [&](const uint32_t k, const uint32_t st = 0u) -> bool {
            unsigned char* stage = ring + st * kStageBytes;
            if (k + 1 < min_blk) {      // every unit of the tile is inside its record range: no range tests
                fetch_block_interior<SVT_STREAM_AUX>(k, src_base, rec_bytes, stage);
                return true;
            } else if (kEdgeAux != SVT_STREAM_AUX && k + 1 >= min_blk) {
                // a unit's last line is shared with its successor's first line.  In the workgroup's LAST tile that successor
                // -- another unit of this workgroup -- has read its first line already (at the start of its own tile): this
                // is the line's second and last use and need not go back into L2 (FETCH_SIZE -10 MB per launch, same time).
                // (Deciding it per line from the neighbour's tile -- also for first lines -- saves 25 MB but costs 1.7 %
                // of the time: sixteen instead of eight first-block instructions, eight more shuffles per tile.)
                constexpr bool kLastUseInLastTile = R >= 2;
                if (kLastUseInLastTile && r == R - 1) fetch_block_tail_exact<SVT_STREAM_AUX, SVT_STREAM_AUX>(k, src_base, src_end, rec_bytes, stage);
                else fetch_block_tail_exact<SVT_STREAM_AUX, kEdgeAux>(k, src_base, src_end, rec_bytes, stage);   // (only the blocks that hold a last line take the edge policy)
            } else
                fetch_block<SVT_STREAM_AUX, false>(k, src_base, src_first, src_end, rec_bytes, stage);
            return false;
        };

        LaneCtx c{};   // kGeneral
        c.is_del = U.svtype == SVT_SVTYPE_DEL;
        c.del16 = c.is_del ? 16u : 0u;
        c.var_length = U.var_length;
        c.pos_delta_d = (double)U.pos_delta;
        StreamCtx sc;  // kSingleLds
        {
            const bool small_del = c.is_del && (c.pos_delta_d < a.lib0.sd2);  // classic.py:339,383
            sc.fmask = small_del ? 0u : 7u;
            sc.kmin = (uint32_t)a.lib0.key_min;
            sc.nb = a.lib0.n_bins;
            sc.sub2 = c.is_del ? (uint32_t)U.var_length + (uint32_t)a.lib0.key_min : 0x80000000u;
            sc.hist_at = kSBins + a.total_bins * 2u;
            sc.wt0 = kSWtab + c.del16 * 8u;
            sc.wt1 = sc.wt0 + 8u * 8u;
            sc.wh0 = kSWhi + c.del16 * 4u;
        }
        if (MODE == kMultiLds) {
            // a window of ONE library (the usual sample) takes the one-library consumer with that library's constants
            const WinLib w0 = reinterpret_cast<const WinLib*>(smem + a.lds_winlibs)[0];
            const bool small_del = c.is_del && (c.pos_delta_d < w0.sd2);
            sc.fmask = small_del ? 0u : 7u;
            sc.kmin = w0.kmin;
            sc.nb = w0.nb;
            sc.sub2 = c.is_del ? (uint32_t)U.var_length + w0.kmin : 0x80000000u;
            sc.hist_at = w0.hist_at;
        }
        WindowCtx wc;  // kMultiLds
        wc.lib_lo = wd.lib_lo;
        wc.lib_last = wd.lib_cnt - 1u;
        wc.winlibs_at = a.lds_winlibs;
        wc.vl_or_never = c.is_del ? (uint32_t)U.var_length : 0x80000000u;
        wc.wt0 = sc.wt0;
        wc.wt1 = sc.wt1;
        wc.wh0 = sc.wh0;
        wc.is_del = c.is_del;
        wc.gated = 0u;
        if (MODE == kMultiLds && c.is_del) {   // (the host keeps windows of more than 32 libraries out of this mode)
            for (uint32_t l = 0; l < wd.lib_cnt; ++l)
                wc.gated |= (c.pos_delta_d < reinterpret_cast<const WinLib*>(smem + a.lds_winlibs)[l].sd2 ? 1u : 0u) << l;
        }
        wc.gated = ~wc.gated;      // (read as `open` bits by record_window)
        Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

        // EDGE = false: every lane's eight records of this block are its own
        auto consume = [&](const u32x4 (&w)[8], const uint32_t k, auto edge, auto window_kind, auto continuations) {
            constexpr bool EDGE = decltype(edge)::value;
            constexpr bool CONT = decltype(continuations)::value;   // sso: some record of the block may continue a fragment
            constexpr int KIND = decltype(window_kind)::value;   // kMultiLds: 1 = the window holds one library, 0 = several
            auto is_mine = [&](const int j) -> bool {
                const uint32_t idx = k * kBlockRecords + (uint32_t)j;
                return !EDGE || (idx >= head && idx < last);
            };
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // keep the look-ups of the second half of the block from being hoisted over the first half: eight
                // records' worth of live table values would not fit the register budget of three waves per SIMD
                if (j == SVT_STREAM_SPLIT) __builtin_amdgcn_sched_barrier(0);
                const bool mine = is_mine(j);
                const u32x4 wj = w[j];
                if (mine) check.see(wj, lib_key);     // (slots that are not this lane's were not fetched)
                if (MODE == kSingleLds) {
                    record_single<SSO, EDGE, CONT>(wj, mine, sc, acc);
                } else if (MODE == kMultiLds) {
                    if (KIND == 1) record_single<SSO, EDGE, CONT>(wj, mine, sc, acc);
                    else record_window<SSO, EDGE, CONT>(wj, mine, wc, acc, check);
                } else {
                    const uint32_t wy = mine ? wj.y : 0u, wz = mine ? wj.z : 0u;   // MAPQ 0 everywhere: adds +0.0
                    weight_evidence<SSO>(wy >> 16 | (wz << 16), wz >> 16, (wj.w & SVT_REC_CONTINUATION) != 0, t, acc);
                    // (a library index beyond the batch's is reported through *err)
                    pair_evidence(wj.x, wy & 0xffffu, wj.w & 7u, min(SVT_REC_LIB(wj.w), a.n_libs - 1u), t, c, acc);
                }
            }
        };

        // kL10Ring: the log10 table of the epilogue goes through the wave's ring once the tile's last block has left it
        auto l10_into_ring = [&]() {
            const char* l10_bytes = reinterpret_cast<const char*>(a.l10);
            for (uint32_t off = 0; off < a.l10_lds_entries * 8u; off += 1024u)
                __builtin_amdgcn_global_load_lds(l10_bytes + off + lane * 16u, (lds_void_ptr)(ring + off), 16, 0, 0);
        };
        if (max_blk) {
            fetch_first();
            u32x4 w[8];
            bool exact_behind = false;   // two stages: is the group issued last -- the one behind the block we wait for -- exactly 8 instructions?
            if (kStreamDepth == 2 && max_blk > 1) exact_behind = fetch(1, 1);
#pragma unroll 1
            for (uint32_t k = 0; k < max_blk; ++k) {
                const uint32_t stage_off = kStreamDepth == 2 ? (k & 1u) * kStageBytes : 0u;
                if (kStreamDepth == 2) {
                    // block k sits in stage k & 1; the group behind it (block k + 1) may stay in flight
                    const bool pend = k + 1 < max_blk && exact_behind;
                    if (k & 1u) {
                        if (pend) read_block<kStageBytes, 8>(lane_block, sw16, w);
                        else read_block<kStageBytes, 0>(lane_block, sw16, w);
                    } else {
                        if (pend) read_block<0, 8>(lane_block, sw16, w);
                        else read_block<0, 0>(lane_block, sw16, w);
                    }
                } else {
                    read_block(lane_block, sw16, w);
                }
                // the block has left its stage for VGPRs: the next fetch into that stage can go out
                auto refill = [&]() {
                    if (kStreamDepth == 2) {
                        if (k + 2 < max_blk) exact_behind = fetch(k + 2, k & 1u);
                    } else if (k + 1 < max_blk) fetch(k + 1);
                    else if (a.l10_where == kL10Ring) l10_into_ring();   // the tile's last block has left the ring: the copy lands while it is summed
                };
                const uint32_t k8 = k * kBlockRecords;
                const uint32_t neutral_w = MODE == kMultiLds ? wd.lib_lo << SVT_REC_LIB_SHIFT : 0u;
                if (__any(k8 < head || k8 + kBlockRecords > last)) {
                    // A slot outside the unit (the neighbours' records in its first / last line, everything past the end of a
                    // shorter unit; not fetched: it holds older bytes) becomes the neutral record: MAPQ 0 everywhere adds +0.0
                    // to every sum (prob_mapq(0) == +0.0), no flag, span 0, the window's first library -- it passes the record
                    // contract and, for the fragment-local sums of the singlesample association, only ever sits in front of the
                    // unit's first record (all sums still 0) or behind its last one (it does the final flush early).
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool mine = k8 + (uint32_t)j - head < n_rec;   // head <= idx < last, unsigned
                        w[j].x = mine ? w[j].x : 0u;
                        w[j].y = mine ? w[j].y : 0u;
                        w[j].z = mine ? w[j].z : 0u;
                        w[j].w = mine ? w[j].w : neutral_w;
                    }
                }
                // sso: a block in which no lane holds a continuation record (nearly all of them: a fragment with a second
                // split candidate of one kind is rare) takes the select-free form of the fragment-local sums.
                bool has_cont = false;
                if (SSO && MODE != kGeneral)
                    has_cont = __any(((w[0].w | w[1].w | w[2].w | w[3].w | w[4].w | w[5].w | w[6].w | w[7].w) & SVT_REC_CONTINUATION) != 0u);
                if (SSO && MODE != kGeneral && has_cont) {
                    // The rare block with a continuation record goes through ONE rolled-up general consumer that takes its records
                    // from the ring again, one at a time (the next fetch waits for it).  An unrolled second consumer beside the
                    // fast one is what made the compiler hoist both consumers' common address arithmetic in front of the branch:
                    // 32 VGPRs alive across every block, three waves per SIMD instead of four.
#pragma unroll 1
                    for (uint32_t j = 0; j < 8u; ++j) {
                        u32x4 wj = *reinterpret_cast<lds_cu32x4*>((size_t)(lane_block + stage_off + ((j << 4) ^ sw16)));
                        const bool mine = k8 + j - head < n_rec;
                        wj.x = mine ? wj.x : 0u;
                        wj.y = mine ? wj.y : 0u;
                        wj.z = mine ? wj.z : 0u;
                        wj.w = mine ? wj.w : neutral_w;
                        check.see(wj, lib_key);
                        if (MODE == kSingleLds || WK == 1 || (WK == 0 && !SSO && wd.lib_cnt == 1u)) record_single<SSO, false, true>(wj, true, sc, acc);
                        else record_window<SSO, false, true>(wj, true, wc, acc, check);
                    }
                    refill();
                    continue;
                }
                refill();
                using kind_any = std::integral_constant<int, 0>;
                using kind_one = std::integral_constant<int, 1>;
                auto run = [&](auto edge_tag, auto kind_tag) {
                    if (!SSO || MODE == kGeneral || has_cont) consume(w, k, edge_tag, kind_tag, std::true_type{});
                    else consume(w, k, edge_tag, kind_tag, std::false_type{});
                };
                // a window of ONE library (the usual sample of a joint run) takes the one-library consumer: 7 % fewer instructions
                // per record (classic association only: the singlesample window kernel spills with both consumers)
                if (MODE == kMultiLds && WK != 2 && (WK == 1 || (!SSO && wd.lib_cnt == 1u))) run(std::false_type{}, kind_one{});   // (workgroup-uniform)
                else run(std::false_type{}, kind_any{});
            }
        }
        if (SSO) {  // flush the last fragment (singlesample.py:370-372)
            acc.ref_seq += acc.l_ref_seq;
            acc.alt_seq += acc.l_alt_seq;
            acc.alt_clip += acc.l_alt_clip;
        }

        // ---- epilogue: the log10 table of log_choose sits beside the tables, or goes through the (now idle) ring
        const double* lds_l10 = reinterpret_cast<const double*>(a.l10_where == kL10Shared ? smem + a.lds_l10 : ring);
        if (a.l10_where == kL10Ring) {
            if (!max_blk || kStreamDepth == 2) l10_into_ring();   // (an empty tile never entered the loop; two stages: copied here)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        uint4 piece[8];
        unit_epilogue(acc, (uint32_t)U.svtype, (uint32_t)U.flags, a.c, lds_l10, a.l10, a.l10_lds_entries, piece);

        // where the record goes: the unit's own index, or (svt_batch_result_order) the site-major index of a sample-major unit
        uint32_t unit_out = unit;
        if (a.out_samples > 1u && unit != kPadUnit) {
            const uint32_t sample = unit / a.out_sites;
            unit_out = (unit - sample * a.out_sites) * a.out_samples + sample;
        }
        {
#if SVT_STORE_DIRECT
            store_results_through_ring(ring, piece, unit_out, lane, a.out);
#else
            // svt_result96: GL, SQ, tallies, QR, QA | GQ, GT, unit -- pieces 0-4 as they are, piece 5 = {GQ, GT, unit, 0}; the tile's 64
            // records go to the tile's own 6 KB of the result buffer, in the tile's (length-sorted) order
            uint32_t tile_slot = 0xFFFFFFFFu;
            if (a.result96) {
                piece[5] = make_uint4(piece[5].x, piece[7].y, unit_out, 0u);   // (a padding lane: unit_out == kPadUnit == SVT_NO_UNIT)
                tile_slot = a.slot_begin + wg_index * kUnitsPerWg + ((uint32_t)r * kWavesPerBlock + ((r & 1) ? (uint32_t)kWavesPerBlock - 1u - wave : wave)) * kWave;
            }
            store_result_records_through_ring(ring, piece, unit_out, lane, reinterpret_cast<unsigned char*>(a.out), a.result96 ? 6u : 8u, tile_slot);
#endif
        }
    }
    const uint32_t bad = check.bits(MODE == kMultiLds ? wd.lib_cnt : a.n_libs);
    if (bad) atomicOr(a.err, bad);
}

}  // namespace svt

#endif  // SVT_STREAM_KERNEL_H
