// svtyper_hip.hip -- MI355X (gfx950 / CDNA4) implementation of the SVTyper likelihood
// hot path behind the C ABI of include/svtyper_hip.h.
//
// Reference lines restated on the device (paths relative to the reference checkout):
//   per-fragment tallies        svtyper/classic.py:296-408, svtyper/singlesample.py:246-353
//   prob_mapq                   svtyper/utils.py:74-75          (256-entry LUT in LDS)
//   p_concordant                svtyper/parsers.py:861-882      (histogram + threshold table in LDS)
//   zeroing rules               svtyper/classic.py:425-435
//   QR/QA + counts              svtyper/classic.py:442-444,455-465
//   log_choose / bayes_gt       svtyper/statistics.py:9-37      (log10 table in LDS, same O(k) loop)
//   GT/GQ/SQ decision           svtyper/classic.py:446,473-495
//
// Design (DESIGN.md has the long form):
//   * one (breakpoint, sample) unit per lane, 64 units per wave ("tile"): the five tallies
//     are sequential binary64 sums in record order, exactly as CPython evaluates them, so the
//     truncated integer counts are bit-exact.  No FMA contraction (-ffp-contract=off).
//   * records are re-tiled once per batch into lane-interleaved tiles: row j of a tile holds
//     the j-th 16 bytes of its 64 units back to back, so every wave-level load is one contiguous
//     1 KiB global_load_dwordx4.  Units are sorted by length inside 4096-unit chunks so that the
//     zero-padding of a tile stays small.  Two device layouts:
//       - split (default): a unit's evidence becomes two sparse 8-byte streams -- pair entries
//         (only fragments with a straddle bit) and weight entries (only fragments with a non-zero
//         gated MAPQ); entries that could only add +0.0 are dropped, which the reference's sums
//         cannot observe;
//       - dense (SVT_FLAG_DENSE_LAYOUT): the canonical 16-byte records as they are.
//   * all look-up tables (prob_mapq, insert-size histogram + p_concordant thresholds, log10,
//     paired-end decision weights) are built on the host with the same libm CPython uses and
//     staged in LDS per workgroup.
//   * HBM-bound byte/integer/fp64 streaming: no MFMA anywhere (nothing here is a contraction).
//
// There is no CPU fallback in this file.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/svtyper_hip.h"

#ifndef SVT_GROUP
#define SVT_GROUP 4   // rows fetched per look-ahead group (dense layout)
#endif
#ifndef SVT_GROUP_A
#define SVT_GROUP_A 2 // split layout, pair-entry rows
#endif
#ifndef SVT_GROUP_B
#define SVT_GROUP_B 2 // split layout, weight-entry rows
#endif
#ifndef SVT_MIN_WAVES
#define SVT_MIN_WAVES 1
#endif
#ifndef SVT_CHUNK
#define SVT_CHUNK 16384
#endif

namespace {

// ------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------
constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kWavesPerBlock = 4;    // 256-thread workgroups
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr uint32_t kChunkUnits = SVT_CHUNK;  // sort window (units)
constexpr uint32_t kPadUnit = 0xFFFFFFFFu;
constexpr uint32_t kMaxLdsTableBytes = 64 * 1024;  // hist+thr budget before falling back to HBM/L2 tables
constexpr uint32_t kMaxL10Lds = 4096;              // log10 table entries kept in LDS (32 KiB)
constexpr uint32_t kTailPadRows = 16;   // look-ahead loads may run this far past a tile (>= 2 * group)

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return fail(SVT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)
#define SVT_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != SVT_OK) return _rc; \
    } while (0)

// ------------------------------------------------------------------------------------------
// device-side structures
// ------------------------------------------------------------------------------------------
struct LibDesc {          // 32 B, one per library
    uint32_t tab_off;     // offset of this library's bins inside hist[] / thr[] (each library
                          // owns n_bins + 1 entries; the last one is the out-of-range sentinel)
    int32_t key_min;
    uint32_t n_bins;
    uint32_t pad;
    double v_nondel;      // lib.mean + lib.sd * 3   (parsers.py:873-875)
    double sd2;           // 2 * lib.sd              (classic.py:339)
};

struct LaneHdr {          // 16 B, one per tile lane
    int32_t var_length;
    int32_t pos_delta;
    uint32_t unit;        // original unit index, kPadUnit for padding lanes
    uint32_t packed;      // svtype | flags << 8 | sample << 16
};

// One 64-unit tile.  Dense layout: rows_a rows of 16-byte records at base_a (rows_b == 0).
// Split layout: rows_a rows of pair entries at base_a, rows_b rows of weight entries at base_b
// (each 16-byte row slot of a lane holds two consecutive 8-byte entries).
struct TileDesc {         // 32 B, stored in dispatch (longest-first) order
    uint64_t base_a;
    uint64_t base_b;
    uint32_t rows_a;
    uint32_t rows_b;
    uint32_t lane_base;   // first LaneHdr of the tile
    uint32_t pad;
};

struct GtConsts {
    double lgp[2][3];     // [is_dup][genotype] log(p)/log(10)      (statistics.py:33-35)
    double lg1p[2][3];    // [is_dup][genotype] log(1-p)/log(10)
    double ln10;          // log(10.0)
    double x_uflow;       // smallest x with libm pow(10.0, x) > 0
    double split_weight;
    double disc_weight;
};

// Paired-end decision table (classic.py:359-405), 32 entries of {w_alt, w_ref}:
//   index = alt_straddle | ref_straddle_A << 1 | ref_straddle_B << 2 | p_concordant << 3 | is_DEL << 4
//   alt_span += (pmA * pmB) * w_alt      w_alt in {0, 1}
//   ref_span += (pmA * pmB) * w_ref      w_ref in {0, 0.5, 1}     ((A + B) * p / 2)
// Multiplying a finite non-negative binary64 by 0, 0.5 or 1 is exact, so this is the reference's
// arithmetic with the branch structure moved into a lookup.
struct PairWeights { double w_alt, w_ref; };

enum LibMode : int {
    kSingleLds = 0,  // one library, descriptor in SGPRs, tables in LDS, 32-bit index math
    kMultiLds = 1,   // several libraries, descriptors + tables in LDS, 32-bit index math
    kGeneral = 2     // any geometry: 64-bit index math, exact float Counter key, tables in HBM/L2
};

// Library window of one workgroup (its 4 tiles): the descriptors [lib_lo, lib_lo + lib_cnt) and the
// histogram/threshold bins [bin_lo, bin_lo + bin_cnt) are the only ones its records can reference, so
// only they are staged in LDS (kMultiLds).  Units are sorted by library first, so a window normally
// holds the 1..3 libraries of one sample.
struct WgDesc {
    uint32_t lib_lo, lib_cnt, bin_lo, bin_cnt;
};

struct KernelArgs {
    const uint4* tiled;
    const TileDesc* tiles;
    const WgDesc* wg;          // one per workgroup (kMultiLds)
    const LaneHdr* hdr;
    const double* pm;          // 256
    const double* l10;         // n_l10
    const LibDesc* libs;       // n_libs
    const uint32_t* hist;      // total_bins (sentinels included)
    const int32_t* thr;        // total_bins
    const PairWeights* wtab;   // 32
    uint32_t n_l10;
    uint32_t n_libs;
    uint32_t total_bins;
    uint32_t n_tiles;
    uint32_t l10_in_lds;
    uint32_t lds_libs;         // LDS capacity in library descriptors (largest window)
    uint32_t lds_bins;         // LDS capacity in histogram bins (largest window)
    uint32_t pad0;
    uint64_t n_units;
    svt_result* out;           // [n_units]
    LibDesc lib0;              // copy of libs[0] (kSingleLds)
    GtConsts c;
};

// ------------------------------------------------------------------------------------------
// evidence arithmetic shared by both layouts
// ------------------------------------------------------------------------------------------
struct Tables {
    const double* pm;          // LDS
    const PairWeights* wtab;   // LDS
    const LibDesc* libs;       // LDS
    const uint32_t* hist;      // LDS (kGeneral: global)
    const int32_t* thr;
};

struct Acc {
    double ref_seq, alt_seq, alt_clip, ref_span, alt_span;
    double l_ref_seq, l_alt_seq, l_alt_clip;  // sso fragment-local sums
};

// per-lane constants of the unit, hoisted out of the record loop
struct LaneCtx {
    uint32_t del16;       // is_DEL ? 16 : 0 (decision-table index bit)
    uint32_t fmask;       // kSingleLds: straddle-bit mask with the small-DEL gate applied
    uint32_t kmin;        // kSingleLds: (uint32) key_min
    uint32_t nb;          // kSingleLds: n_bins (== sentinel index)
    uint32_t sub2;        // kSingleLds: DEL ? var_length + key_min : 0x80000000 (never in range)
    uint32_t lib_lo;      // kMultiLds: first library / first bin staged by this workgroup
    uint32_t bin_lo;
    int32_t var_length;
    double pos_delta_d;
    bool is_del;
};

// Split-read / reference-read weights of one fragment record (classic.py:306-328).  Every add is
// unconditional: gated-off evidence arrives as MAPQ 0, whose weight prob_mapq(0) is exactly +0.0,
// and x + 0.0 == x bit-for-bit for these non-negative sums.
//   wa = rs_a | rs_b << 8 | seq_l << 16 | seq_r << 24,  wb = clip_l | clip_r << 8
template <bool SSO>
__device__ __forceinline__ void weight_evidence(const uint32_t wa, const uint32_t wb, const bool cont,
                                                const Tables& t, Acc& a)
{
    const double rs_a = t.pm[wa & 0xffu];
    const double rs_b = t.pm[(wa >> 8) & 0xffu];
    const double sq_l = t.pm[(wa >> 16) & 0xffu];
    const double sq_r = t.pm[wa >> 24];
    const double cl_l = t.pm[wb & 0xffu];
    const double cl_r = t.pm[(wb >> 8) & 0xffu];
    // p_alt = (pm(left) * L + pm(right) * R) / 2.0   (classic.py:324)
    const double p_seq = (sq_l + sq_r) * 0.5;
    const double p_clip = (cl_l + cl_r) * 0.5;
    if (SSO) {
        // singlesample.py:246-276,367-372: per-fragment sums starting from 0, added to the site
        // totals when the next fragment starts
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
}

// Paired-end evidence of one fragment (classic.py:339-408).
//   o = ospan_len, mq = mapq_a | mapq_b << 8, f3 = alt | refA << 1 | refB << 2, lib = library index
template <int MODE>
__device__ __forceinline__ void pair_evidence(const uint32_t o, const uint32_t mq, uint32_t f3,
                                              const uint32_t lib_idx, const Tables& t, const LaneCtx& c, Acc& a)
{
    const double pm_a = t.pm[mq & 0xffu];
    const double pm_b = t.pm[(mq >> 8) & 0xffu];

    // p_concordant (parsers.py:861-882) as an integer test: with d1 = hist[o]/N fixed, the
    // reference's binary64 expression d1*0.95/(0.95*d1 + 0.05*d2) > 0.5 is monotone in
    // h2 = hist[o - v]; thr[o] is the largest h2 for which it still holds (found on the host with
    // the reference's own expression), -1 where hist[o] == 0 (p == 0 or ZeroDivisionError).
    int32_t thr1;
    uint32_t h2;
    if (MODE == kSingleLds) {
        f3 &= c.fmask;                                  // small-DEL gate (classic.py:339,383)
        const uint32_t i1 = min(o - c.kmin, c.nb);      // out of range -> sentinel (thr -1)
        const uint32_t i2 = min(o - c.sub2, c.nb);      // out of range -> sentinel (hist 0)
        thr1 = t.thr[i1];
        h2 = t.hist[i2];
    } else if (MODE == kMultiLds) {
        const LibDesc lib = t.libs[lib_idx - c.lib_lo];
        const bool small_del = c.is_del && (c.pos_delta_d < lib.sd2);
        f3 = small_del ? 0u : f3;
        const uint32_t kmin = (uint32_t)lib.key_min;
        const uint32_t sub2 = c.is_del ? (uint32_t)c.var_length + kmin : 0x80000000u;
        const uint32_t i1 = min(o - kmin, lib.n_bins);
        const uint32_t i2 = min(o - sub2, lib.n_bins);
        const uint32_t base = lib.tab_off - c.bin_lo;
        thr1 = t.thr[base + i1];
        h2 = t.hist[base + i2];
    } else {
        const LibDesc lib = t.libs[lib_idx];
        const bool small_del = c.is_del && (c.pos_delta_d < lib.sd2);
        f3 = small_del ? 0u : f3;
        const int64_t i1 = (int64_t)(int32_t)o - (int64_t)lib.key_min;
        const bool in1 = (uint64_t)i1 < (uint64_t)lib.n_bins;
        thr1 = t.thr[lib.tab_off + (in1 ? (uint32_t)i1 : lib.n_bins)];
        int64_t key2;
        bool ok2 = true;
        if (c.is_del) {
            key2 = (int64_t)(int32_t)o - (int64_t)c.var_length;
        } else {
            // var_length is None: the Counter key is the FLOAT o - (mean + 3 sd); it only matches
            // an integer key when it is integral (parsers.py:874-878)
            const double kf = (double)(int32_t)o - lib.v_nondel;
            ok2 = (kf == floor(kf)) && (fabs(kf) < 4.0e9);
            key2 = ok2 ? (int64_t)kf : 0;
        }
        const int64_t i2 = key2 - (int64_t)lib.key_min;
        const bool in2 = ok2 && ((uint64_t)i2 < (uint64_t)lib.n_bins);
        h2 = t.hist[lib.tab_off + (in2 ? (uint32_t)i2 : lib.n_bins)];
    }
    const bool p_conc = (int32_t)h2 <= thr1;
    const PairWeights pw = t.wtab[f3 | (p_conc ? 8u : 0u) | c.del16];
    const double pp = pm_a * pm_b;
    a.alt_span += pp * pw.w_alt;
    a.ref_span += pp * pw.w_ref;
}

__device__ __forceinline__ double log_choose_dev(const double* __restrict__ l10, int32_t n, int32_t k)
{
    // statistics.py:9-20 -- same loop, log(i)/log(10) from the host-built table
    double r = 0.0;
    if (k * 2 > n) k = n - k;
    for (int32_t d = 1; d <= k; ++d) {
        r += l10[n];
        r -= l10[d];
        n -= 1;
    }
    return r;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// streaming read of one 16-byte row slot (read exactly once per pass): non-temporal
__device__ __forceinline__ uint4 ld_stream(const uint4* __restrict__ p)
{
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint4 pack2d(double x, double y)
{
    const uint64_t a = (uint64_t)__double_as_longlong(x), b = (uint64_t)__double_as_longlong(y);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}

// Stream `rows` row slots of one lane, SVT_GROUP at a time, one group ahead of the group being
// consumed (the tiled buffer carries kTailPadRows rows of slack, so the look-ahead never leaves
// the allocation).
template <int G, typename F>
__device__ __forceinline__ void stream_rows(const uint4* __restrict__ p, const uint32_t rows, F&& consume)
{
    uint4 cur[G], nxt[G];
#pragma unroll
    for (int k = 0; k < G; ++k) cur[k] = ld_stream(p + k * kWave);
    uint32_t j = 0;
    for (; j + G <= rows; j += G) {
        const uint4* __restrict__ q = p + (uint64_t)(j + G) * kWave;
#pragma unroll
        for (int k = 0; k < G; ++k) nxt[k] = ld_stream(q + k * kWave);
#pragma unroll
        for (int k = 0; k < G; ++k) consume(cur[k]);
#pragma unroll
        for (int k = 0; k < G; ++k) cur[k] = nxt[k];
    }
    const uint32_t rem = rows - j;  // wave-uniform
#pragma unroll
    for (int k = 0; k < G - 1; ++k)
        if ((uint32_t)k < rem) consume(cur[k]);
}

// ------------------------------------------------------------------------------------------
// genotype kernel
// ------------------------------------------------------------------------------------------
template <bool SSO, int MODE, bool SPLIT>
__global__ __launch_bounds__(kBlock, SVT_MIN_WAVES) void svt_genotype_kernel(const KernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS layout: pm[256] | wtab[32] | l10[n_l10 (even)] | libs[n_libs] | hist[total_bins] | thr[total_bins]
    double* s_pm = reinterpret_cast<double*>(smem);
    PairWeights* s_wtab = reinterpret_cast<PairWeights*>(s_pm + 256);
    double* s_l10 = reinterpret_cast<double*>(s_wtab + 32);
    const uint32_t n_l10_lds = a.l10_in_lds ? ((a.n_l10 + 1u) & ~1u) : 0u;
    LibDesc* s_lib = reinterpret_cast<LibDesc*>(s_l10 + n_l10_lds);
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_lib + a.lds_libs);
    int32_t* s_thr = reinterpret_cast<int32_t*>(s_hist + a.lds_bins);
    // library window of this workgroup (everything when the tables of the whole batch fit)
    WgDesc wd = {0u, a.n_libs, 0u, MODE != kGeneral ? a.total_bins : 0u};
    if (MODE == kMultiLds) wd = a.wg[blockIdx.x];

    // ---- stage the tables in LDS (they are L2-resident after the first workgroups)
    for (uint32_t i = threadIdx.x; i < 256; i += kBlock) s_pm[i] = a.pm[i];
    if (threadIdx.x < 32) s_wtab[threadIdx.x] = a.wtab[threadIdx.x];
    if (a.l10_in_lds)
        for (uint32_t i = threadIdx.x; i < a.n_l10; i += kBlock) s_l10[i] = a.l10[i];
    for (uint32_t i = threadIdx.x; i < wd.lib_cnt * (uint32_t)(sizeof(LibDesc) / 8); i += kBlock)
        reinterpret_cast<uint64_t*>(s_lib)[i] =
            reinterpret_cast<const uint64_t*>(a.libs + wd.lib_lo)[i];
    if (MODE != kGeneral) {
        for (uint32_t i = threadIdx.x; i < wd.bin_cnt; i += kBlock) {
            s_hist[i] = a.hist[wd.bin_lo + i];
            s_thr[i] = a.thr[wd.bin_lo + i];
        }
    }
    __syncthreads();

    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x % kWave;
    const uint32_t tile_idx = blockIdx.x * kWavesPerBlock + wave;
    if (tile_idx >= a.n_tiles) return;

    const TileDesc td = a.tiles[tile_idx];
    if (td.lane_base == kPadUnit) return;   // padding of the last workgroup
    const LaneHdr h = a.hdr[td.lane_base + lane];
    const uint32_t svtype = h.packed & 0xffu;
    const uint32_t uflags = (h.packed >> 8) & 0xffu;

    Tables t;
    t.pm = s_pm;
    t.wtab = s_wtab;
    t.libs = s_lib;
    t.hist = MODE != kGeneral ? s_hist : a.hist;
    t.thr = MODE != kGeneral ? s_thr : a.thr;

    LaneCtx c;
    c.is_del = svtype == SVT_SVTYPE_DEL;
    c.del16 = c.is_del ? 16u : 0u;
    c.var_length = h.var_length;
    c.pos_delta_d = (double)h.pos_delta;
    c.lib_lo = wd.lib_lo;
    c.bin_lo = wd.bin_lo;
    {
        const bool small_del = c.is_del && (c.pos_delta_d < a.lib0.sd2);  // classic.py:339,383
        c.fmask = small_del ? 0u : 7u;
        c.kmin = (uint32_t)a.lib0.key_min;
        c.nb = a.lib0.n_bins;
        c.sub2 = c.is_del ? (uint32_t)h.var_length + (uint32_t)a.lib0.key_min : 0x80000000u;
    }

    Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

    // ---- stream the tile: row j is one contiguous 1 KiB line for the wave
    if (SPLIT) {
        // pair entries: x = ospan_len, y = mapq_a | mapq_b << 8 | f3 << 16 | lib << 24
        stream_rows<SVT_GROUP_A>(a.tiled + td.base_a + lane, td.rows_a, [&](const uint4 w) {
            pair_evidence<MODE>(w.x, w.y & 0xffffu, (w.y >> 16) & 7u, w.y >> 24, t, c, acc);
            pair_evidence<MODE>(w.z, w.w & 0xffffu, (w.w >> 16) & 7u, w.w >> 24, t, c, acc);
        });
        // weight entries: x = rs_a | rs_b << 8 | seq_l << 16 | seq_r << 24, y = clip_l | clip_r << 8 | cont << 16
        stream_rows<SVT_GROUP_B>(a.tiled + td.base_b + lane, td.rows_b, [&](const uint4 w) {
            weight_evidence<SSO>(w.x, w.y, (w.y & 0x10000u) != 0, t, acc);
            weight_evidence<SSO>(w.z, w.w, (w.w & 0x10000u) != 0, t, acc);
        });
    } else {
        // canonical 16-byte records (include/svtyper_hip.h: svt_record)
        stream_rows<SVT_GROUP>(a.tiled + td.base_a + lane, td.rows_a, [&](const uint4 w) {
            weight_evidence<SSO>(w.y >> 16 | (w.z << 16), w.z >> 16, (w.w & SVT_REC_CONTINUATION) != 0, t, acc);
            pair_evidence<MODE>(w.x, w.y & 0xffffu, w.w & 7u, SVT_REC_LIB(w.w), t, c, acc);
        });
    }
    if (SSO) {  // flush the last fragment (singlesample.py:370-372)
        acc.ref_seq += acc.l_ref_seq;
        acc.alt_seq += acc.l_alt_seq;
        acc.alt_clip += acc.l_alt_clip;
    }

    if (h.unit == kPadUnit) return;

    double ref_seq = acc.ref_seq, alt_seq = acc.alt_seq, alt_clip = acc.alt_clip,
           ref_span = acc.ref_span, alt_span = acc.alt_span;

    // ---- zeroing rules (classic.py:425-435)
    if ((alt_seq + alt_clip) < 0.5 && alt_span >= 1.0) { alt_seq = 0.0; alt_clip = 0.0; ref_seq = 0.0; }
    if (alt_span < 0.5 && (alt_seq + alt_clip) >= 1.0) { alt_span = 0.0; ref_span = 0.0; }
    if (alt_span + alt_seq == 0.0 && alt_clip > 0.0) alt_clip = 0.0;

    int32_t cnt[SVT_N_COUNTS];
#pragma unroll
    for (int i = 0; i < SVT_N_COUNTS; ++i) cnt[i] = 0;
    double gl[3] = {0.0, 0.0, 0.0};
    double sq = 0.0;
    int32_t gt;

    const bool skipped = (uflags & SVT_UNIT_SKIP) != 0;
    const bool evidence = (ref_seq + alt_seq + ref_span + alt_span + alt_clip) > 0.0;  // classic.py:437
    if (skipped) {
        ref_seq = alt_seq = alt_clip = ref_span = alt_span = 0.0;
        gt = SVT_GT_SKIPPED;
        cnt[SVT_CNT_GQ] = -1;
    } else if (!evidence) {
        gt = SVT_GT_BLANK;  // classic.py:496-513
        cnt[SVT_CNT_GQ] = -1;
    } else {
        const int is_dup = svtype == SVT_SVTYPE_DUP;                                  // :439
        const double alt_splitters = alt_seq + alt_clip;                              // :442
        const int32_t QR = (int32_t)(a.c.split_weight * ref_seq) + (int32_t)(a.c.disc_weight * ref_span);      // :443
        const int32_t QA = (int32_t)(a.c.split_weight * alt_splitters) + (int32_t)(a.c.disc_weight * alt_span); // :444
        // bayes_gt (statistics.py:23-37)
        const int32_t total = QR + QA;
        double log_combo;
        if (a.l10_in_lds) log_combo = log_choose_dev(s_l10, total, QA);
        else log_combo = log_choose_dev(a.l10, total, QA);
#pragma unroll
        for (int g = 0; g < 3; ++g)
            gl[g] = (log_combo + (double)QA * a.c.lgp[is_dup][g]) + (double)QR * a.c.lg1p[is_dup][g];

        // stable descending order of (index, value): ties keep the lower index (classic.py:446)
        int best = 0;
        if (gl[1] > gl[best]) best = 1;
        if (gl[2] > gl[best]) best = 2;
        const int r0 = best == 0 ? 1 : 0;
        const int r1 = best == 2 ? 1 : 2;
        const int second = (gl[r1] > gl[r0]) ? r1 : r0;

        cnt[SVT_CNT_QR] = QR;
        cnt[SVT_CNT_QA] = QA;
        cnt[SVT_CNT_DP] = (int32_t)(ref_seq + alt_seq + alt_clip + ref_span + alt_span);  // :455
        cnt[SVT_CNT_RO] = (int32_t)(ref_seq + ref_span);                                  // :456
        cnt[SVT_CNT_AO] = (int32_t)(alt_seq + alt_clip + alt_span);                       // :457
        cnt[SVT_CNT_RS] = (int32_t)ref_seq;
        cnt[SVT_CNT_AS] = (int32_t)alt_seq;
        cnt[SVT_CNT_ASC] = (int32_t)alt_clip;
        cnt[SVT_CNT_RP] = (int32_t)ref_span;
        cnt[SVT_CNT_AP] = (int32_t)alt_span;

        // gt_sum = sum(10**gl) (classic.py:473-478).  Whether it is > 0 is decided against the
        // host libm's own underflow point of pow(10, x), so GT './.' agrees with CPython.
        const double gl_best = gl[best];
        if (gl_best >= a.c.x_uflow) {
            double gt_sum = 0.0;
#pragma unroll
            for (int g = 0; g < 3; ++g) gt_sum += pow(10.0, gl[g]);
            const double gt_sum_log = log(gt_sum) / a.c.ln10;                       // :480
            sq = fabs(-10.0 * (gl[0] - gt_sum_log));                                // :481
            double phred_gq = -10.0 * (gl[second] - gl_best);                       // :482
            if (phred_gq > 200.0) phred_gq = 200.0;
            cnt[SVT_CNT_GQ] = (int32_t)phred_gq;                                    // :483
            gt = best;
        } else {
            cnt[SVT_CNT_GQ] = -1;                                                   // :493-495
            gt = SVT_GT_MISSING;
        }
    }

    // ---- one 128-byte result record per unit = one full L2 line written by one lane: the
    // scatter back to the unit's original position costs no partial-line traffic
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(a.out + h.unit);
    dst[0] = pack2d(gl[0], gl[1]);
    dst[1] = pack2d(gl[2], sq);
    dst[2] = pack2d(ref_seq, alt_seq);
    dst[3] = pack2d(alt_clip, ref_span);
    {
        const uint64_t t4 = (uint64_t)__double_as_longlong(alt_span);
        dst[4] = make_uint4((uint32_t)t4, (uint32_t)(t4 >> 32), (uint32_t)cnt[0], (uint32_t)cnt[1]);
    }
    dst[5] = make_uint4((uint32_t)cnt[2], (uint32_t)cnt[3], (uint32_t)cnt[4], (uint32_t)cnt[5]);
    dst[6] = make_uint4((uint32_t)cnt[6], (uint32_t)cnt[7], (uint32_t)cnt[8], (uint32_t)cnt[9]);
    dst[7] = make_uint4((uint32_t)cnt[10], (uint32_t)gt & 0xffu, 0u, 0u);
}

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

// ------------------------------------------------------------------------------------------
// geometry kernel: one fragment summary per thread -> one canonical evidence record
// (svtyper/parsers.py:785-857, 1122-1215; the walk of svtyper/classic.py:296-396 per fragment)
// ------------------------------------------------------------------------------------------
struct GeomArgs {
    const uint4* frags;          // svt_fragment[n_frags] viewed as 8 x uint4
    const uint32_t* frag_unit;   // unit of every fragment summary
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
    const svt_breakpoint bp = g.bps[g.frag_unit[i]];
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

// ------------------------------------------------------------------------------------------
// bayes_gt seam kernel: one (ref, alt, is_dup) item per thread (statistics.py:9-37)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void svt_bayes_kernel(const int32_t* __restrict__ ref,
                                                           const int32_t* __restrict__ alt,
                                                           const uint8_t* __restrict__ is_dup,
                                                           uint64_t n, const double* __restrict__ l10,
                                                           const GtConsts c, double* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t r = ref[i], a = alt[i];
    const int d = is_dup[i] ? 1 : 0;
    const double log_combo = log_choose_dev(l10, r + a, a);
    double4 o;
    o.x = (log_combo + (double)a * c.lgp[d][0]) + (double)r * c.lg1p[d][0];
    o.y = (log_combo + (double)a * c.lgp[d][1]) + (double)r * c.lg1p[d][1];
    o.z = (log_combo + (double)a * c.lgp[d][2]) + (double)r * c.lg1p[d][2];
    o.w = log_combo;
    reinterpret_cast<double4*>(out)[i] = o;
}

// ------------------------------------------------------------------------------------------
// host-side table construction (same libm calls CPython makes)
// ------------------------------------------------------------------------------------------

// parsers.py:861-882 for counts (h1, h2) of a library with N samples
bool p_concordant_expr(uint32_t h1, uint32_t h2, uint64_t n_total)
{
    const double disc_prior = 0.05;
    const double conc_prior = 1 - disc_prior;
    const double d1 = h1 ? (double)h1 / (double)n_total : 0.0;  // parsers.py:582
    const double d2 = h2 ? (double)h2 / (double)n_total : 0.0;
    const double den = conc_prior * d1 + disc_prior * d2;
    if (den == 0.0) return false;  // ZeroDivisionError -> None -> (None > 0.5) == False
    const double p = d1 * conc_prior / den;
    return p > 0.5;
}

double py_log10(double x) { return std::log(x) / std::log(10.0); }  // math.log(x, 10)

// smallest double x with pow(10.0, x) > 0 under this libm (CPython: 10 ** x)
double find_pow10_underflow()
{
    double lo = -330.0, hi = -300.0;  // pow(10,lo) == 0, pow(10,hi) > 0
    for (int it = 0; it < 200; ++it) {
        double mid = lo + (hi - lo) / 2;
        if (mid == lo || mid == hi) break;
        if (std::pow(10.0, mid) > 0.0) hi = mid; else lo = mid;
    }
    // walk to the exact boundary in ulps
    while (std::pow(10.0, std::nextafter(hi, -INFINITY)) > 0.0) hi = std::nextafter(hi, -INFINITY);
    return hi;
}

void fill_gt_consts(GtConsts& c, double split_weight, double disc_weight)
{
    const double p_alt[2][3] = {{1e-3, 0.5, 0.9}, {1e-2, 0.2, 1 / 3.0}};  // statistics.py:26,28
    for (int d = 0; d < 2; ++d)
        for (int g = 0; g < 3; ++g) {
            c.lgp[d][g] = py_log10(p_alt[d][g]);
            c.lg1p[d][g] = py_log10(1 - p_alt[d][g]);
        }
    c.ln10 = std::log(10.0);
    c.x_uflow = find_pow10_underflow();
    c.split_weight = split_weight;
    c.disc_weight = disc_weight;
}

struct HostTables {
    std::vector<LibDesc> libs;
    std::vector<uint32_t> hist;      // per library: n_bins counts + sentinel 0
    std::vector<int32_t> thr;        // per library: n_bins thresholds + sentinel -1
    std::vector<PairWeights> wtab;   // 32
    std::vector<double> pm;          // 256
    std::vector<double> l10;
    bool fast_geometry = true;       // 32-bit index math + "non-DEL key never integral" valid?
};

int build_tables(const svt_evidence_batch* in, uint64_t max_records_per_unit, HostTables& T)
{
    T.libs.resize(in->n_libs);
    for (uint32_t l = 0; l < in->n_libs; ++l) {
        const svt_library& L = in->libs[l];
        if (!L.hist || L.n_bins == 0) return fail(SVT_ERR_INVALID, "library without histogram");
        if (L.n_bins > (1u << 24)) return fail(SVT_ERR_INVALID, "histogram too wide");
        if (!std::isfinite(L.mean) || !std::isfinite(L.sd)) return fail(SVT_ERR_INVALID, "library moments not finite");
        uint64_t total = 0;
        uint32_t hmax = 0;
        for (uint32_t i = 0; i < L.n_bins; ++i) {
            if (L.hist[i] > 0x7FFFFFFFu) return fail(SVT_ERR_INVALID, "histogram count too large");
            total += L.hist[i];
            hmax = std::max(hmax, L.hist[i]);
        }
        LibDesc d{};
        d.tab_off = (uint32_t)T.hist.size();
        d.key_min = L.key_min;
        d.n_bins = L.n_bins;
        d.v_nondel = L.mean + L.sd * 3;  // parsers.py:873-875
        d.sd2 = 2 * L.sd;                // classic.py:339
        T.libs[l] = d;
        // the fast kernels need |key_min| <= 2^29 and a non-DEL float key o - (mean + 3 sd) that can
        // never round to an integer for o in [0, 2^31)
        if (L.key_min < -(1 << 29) || L.key_min > (1 << 29)) T.fast_geometry = false;
        if (!(std::fabs(d.v_nondel - std::nearbyint(d.v_nondel)) > 4e-6) || !(std::fabs(d.v_nondel) < 1e12))
            T.fast_geometry = false;
        for (uint32_t i = 0; i < L.n_bins; ++i) {
            const uint32_t h1 = L.hist[i];
            T.hist.push_back(h1);
            int32_t t = -1;
            if (h1 > 0 && total > 0 && p_concordant_expr(h1, 0, total)) {
                // largest h2 in [0, hmax] with p > 0.5 (the expression is monotone non-increasing in h2)
                uint32_t lo = 0, hi = hmax;  // invariant: expr(lo) holds
                if (p_concordant_expr(h1, hi, total)) lo = hi;
                else
                    while (hi - lo > 1) {
                        const uint32_t mid = lo + (hi - lo) / 2;
                        if (p_concordant_expr(h1, mid, total)) lo = mid; else hi = mid;
                    }
                t = (int32_t)lo;
            }
            T.thr.push_back(t);
        }
        T.hist.push_back(0);   // out-of-range sentinel: Counter miss -> 0
        T.thr.push_back(-1);   //                        hist[o] == 0 -> never concordant
    }
    // paired-end decision table (see PairWeights)
    T.wtab.resize(32);
    for (int i = 0; i < 32; ++i) {
        const bool alt = i & 1, ra = i & 2, rb = i & 4, pc = i & 8, del = i & 16;
        const bool both = ra && rb, any = ra || rb;
        const bool need = any && (!both || del);                   // classic.py:398-401
        T.wtab[i].w_alt = (alt && !(del && pc)) ? 1.0 : 0.0;       // classic.py:359-377
        T.wtab[i].w_ref = (need && pc) ? (both ? 1.0 : 0.5) : 0.0; // classic.py:402-405
    }
    // log10 table: n = QR + QA <= 2 * (2 * split_weight + disc_weight) * max F
    const double bound = 2.0 * (2.0 * in->split_weight + in->disc_weight) * (double)max_records_per_unit + 4.0;
    if (bound > 64.0 * 1024 * 1024) return fail(SVT_ERR_INVALID, "weights * records too large for the log table");
    T.l10.resize((size_t)bound + 1);
    T.l10[0] = 0.0;  // never read (log_choose only looks up 1..n)
    for (size_t i = 1; i < T.l10.size(); ++i) T.l10[i] = py_log10((double)i);
    T.pm.resize(256);
    for (int q = 0; q < 256; ++q) T.pm[q] = 1.0 - std::pow(10.0, -(double)q / 10.0);  // utils.py:74-75
    return SVT_OK;
}

// ------------------------------------------------------------------------------------------
// host-side tiling
// ------------------------------------------------------------------------------------------
struct Tiling {
    std::vector<TileDesc> tiles;       // storage order
    std::vector<uint32_t> tile_lib_lo, tile_lib_hi;  // library range referenced by each tile
    std::vector<LaneHdr> hdr;
    std::vector<uint64_t> lane_src;
    std::vector<uint32_t> lane_nrec;
    uint64_t slots = 0;                // 16-byte row slots of all tiles
};

unsigned host_threads()
{
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hc ? hc : 1u, 16u));
}

// run fn(i) for i in [0, n) on up to host_threads() threads
template <typename Fn>
void parallel_for(uint64_t n, Fn&& fn)
{
    const unsigned nt = (unsigned)std::min<uint64_t>(host_threads(), n);
    if (nt <= 1) {
        for (uint64_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&, t]() { for (uint64_t i = t; i < n; i += nt) fn(i); });
    for (auto& th : pool) th.join();
}

// Sort units by library and stream length inside chunks, cut into 64-unit tiles.  len_a/len_b are
// the per-unit row counts of the two streams (dense layout: len_a = F, len_b = 0).  Chunks are
// independent and are processed by several host threads.
void build_tiling(const svt_evidence_batch* in, const std::vector<uint32_t>& nrec,
                  const std::vector<uint32_t>& len_a, const std::vector<uint32_t>& len_b,
                  const std::vector<uint4>& scan, Tiling& G)
{
    const uint64_t n = in->n_units;
    const uint64_t n_chunks = (n + kChunkUnits - 1) / kChunkUnits;
    const uint64_t tiles_per_chunk = kChunkUnits / kWave;
    const uint64_t n_tiles = n ? (n_chunks - 1) * tiles_per_chunk +
                                     ((n - (n_chunks - 1) * kChunkUnits) + kWave - 1) / kWave : 0;
    G.tiles.assign(n_tiles, TileDesc{});
    G.tile_lib_lo.assign(n_tiles, 0);
    G.tile_lib_hi.assign(n_tiles, 0);
    G.hdr.assign(n_tiles * kWave, LaneHdr{});
    G.lane_src.assign(n_tiles * kWave, 0);
    G.lane_nrec.assign(n_tiles * kWave, 0);
    parallel_for(n_chunks, [&](uint64_t c) {
        const uint64_t c0 = c * kChunkUnits;
        const uint32_t cn = (uint32_t)std::min<uint64_t>(kChunkUnits, n - c0);
        std::vector<uint32_t> order(cn);
        for (uint32_t i = 0; i < cn; ++i) order[i] = i;
        // by first library (keeps the units of one sample together so a workgroup's library window
        // stays small), then longest first
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const uint32_t lx = scan[c0 + x].z, ly = scan[c0 + y].z;
            if (lx != ly) return lx < ly;
            const uint64_t kx = ((uint64_t)len_a[c0 + x] << 32) | len_b[c0 + x];
            const uint64_t ky = ((uint64_t)len_a[c0 + y] << 32) | len_b[c0 + y];
            return kx > ky;
        });
        for (uint32_t t0 = 0; t0 < cn; t0 += kWave) {
            const uint64_t ti = c * tiles_per_chunk + t0 / kWave;
            TileDesc td{};
            td.lane_base = (uint32_t)(ti * kWave);
            uint32_t lib_lo = 0xffffffffu, lib_hi = 0;
            for (uint32_t l = 0; l < (uint32_t)kWave; ++l) {
                LaneHdr h{};
                h.unit = kPadUnit;
                uint64_t src = 0;
                uint32_t f = 0;
                if (t0 + l < cn) {
                    const uint64_t u = c0 + order[t0 + l];
                    const svt_unit& U = in->units[u];
                    h.var_length = U.var_length;
                    h.pos_delta = U.pos_delta;
                    h.unit = (uint32_t)u;
                    h.packed = (uint32_t)U.svtype | ((uint32_t)U.flags << 8) | ((uint32_t)U.sample << 16);
                    src = in->rec_offset[u];
                    f = nrec[u];
                    td.rows_a = std::max(td.rows_a, len_a[u]);
                    td.rows_b = std::max(td.rows_b, len_b[u]);
                    if (f) {
                        lib_lo = std::min(lib_lo, scan[u].z);
                        lib_hi = std::max(lib_hi, scan[u].w);
                    }
                }
                G.hdr[td.lane_base + l] = h;
                G.lane_src[td.lane_base + l] = src;
                G.lane_nrec[td.lane_base + l] = f;
            }
            G.tiles[ti] = td;
            G.tile_lib_lo[ti] = lib_lo == 0xffffffffu ? 0u : lib_lo;
            G.tile_lib_hi[ti] = lib_lo == 0xffffffffu ? 0u : lib_hi;
        }
    });
    // slot offsets: a tile's pair rows, then its weight rows
    for (TileDesc& td : G.tiles) {
        td.base_a = G.slots;
        td.base_b = G.slots + (uint64_t)td.rows_a * kWave;
        G.slots += (uint64_t)(td.rows_a + td.rows_b) * kWave;
    }
}

// Device scratch for the canonical records of the batch being created: a 1.6 GB hipMalloc costs
// ~100 ms, so the buffer is kept per device between calls (grow-only; svt_trim() releases it).
struct CsrScratchCache {
    static constexpr int kMaxDevices = 64;
    void* ptr[kMaxDevices] = {};
    uint64_t cap[kMaxDevices] = {};
    std::mutex lock;   // held for the whole svt_batch_create of a device-sharing caller
    int acquire(int device, uint64_t bytes, void** out)
    {
        if (device >= kMaxDevices) return fail(SVT_ERR_INVALID, "device index too large for the scratch cache");
        if (cap[device] < bytes) {
            if (ptr[device]) (void)hipFree(ptr[device]);
            ptr[device] = nullptr;
            cap[device] = 0;
            const uint64_t want = bytes + bytes / 8;   // a little slack for the next, slightly larger batch
            HIP_TRY(hipMalloc(&ptr[device], want));
            cap[device] = want;
        }
        *out = ptr[device];
        return SVT_OK;
    }
    void trim()
    {
        std::lock_guard<std::mutex> g(lock);
        for (int d = 0; d < kMaxDevices; ++d)
            if (ptr[d]) {
                (void)hipSetDevice(d);
                (void)hipFree(ptr[d]);
                ptr[d] = nullptr;
                cap[d] = 0;
            }
    }
};
CsrScratchCache g_csr_cache;

// Pinned staging ring shared by all batches of the process (allocated on first use, per device
// context of the first caller; pinned host memory is usable from every device).
struct StagingRing {
    static constexpr uint64_t kPiece = 64ull << 20;
    static constexpr int kSlots = 3;
    void* buf[kSlots] = {nullptr, nullptr, nullptr};
    std::mutex lock;
    int ensure()
    {
        for (int i = 0; i < kSlots; ++i)
            if (!buf[i] && hipHostMalloc(&buf[i], kPiece, hipHostMallocDefault) != hipSuccess)
                return fail(SVT_ERR_HIP, "hipHostMalloc of the pinned staging ring failed");
        return SVT_OK;
    }
};
StagingRing g_ring;

// Host -> device copy of a large pageable buffer through the pinned ring: a few host threads fill
// one piece while the previous piece is on the wire (a first hipMemcpy of pageable memory stages at
// ~13 GB/s on this platform; pinned pieces move at ~56 GB/s, tools/h2d_probe.hip).
int h2d_staged(void* dst, const void* src, uint64_t bytes, hipStream_t stream)
{
    if (bytes < (16ull << 20)) {
        if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        return SVT_OK;
    }
    std::lock_guard<std::mutex> guard(g_ring.lock);
    SVT_TRY(g_ring.ensure());
    hipEvent_t done[StagingRing::kSlots] = {nullptr, nullptr, nullptr};
    int rc = SVT_OK;
    for (int i = 0; i < StagingRing::kSlots && rc == SVT_OK; ++i)
        if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) rc = fail(SVT_ERR_HIP, "hipEventCreate");
    const unsigned nt = std::min(host_threads(), 6u);   // 4-8 threads saturate the host copy
    uint64_t off = 0;
    for (int slot = 0; rc == SVT_OK && off < bytes; slot = (slot + 1) % StagingRing::kSlots) {
        const uint64_t len = std::min(StagingRing::kPiece, bytes - off);
        if (hipEventSynchronize(done[slot]) != hipSuccess) { rc = fail(SVT_ERR_HIP, "staging event"); break; }
        const char* s0 = static_cast<const char*>(src) + off;
        char* p0 = static_cast<char*>(g_ring.buf[slot]);
        const uint64_t part = ((len + nt - 1) / nt + 4095) & ~uint64_t(4095);
        parallel_for(nt, [&](uint64_t t) {
            const uint64_t lo = t * part, hi = std::min(len, lo + part);
            if (lo < hi) std::memcpy(p0 + lo, s0 + lo, hi - lo);
        });
        if (hipMemcpyAsync(static_cast<char*>(dst) + off, p0, len, hipMemcpyHostToDevice, stream) != hipSuccess ||
            hipEventRecord(done[slot], stream) != hipSuccess) { rc = fail(SVT_ERR_HIP, "staged hipMemcpyAsync"); break; }
        off += len;
    }
    (void)hipStreamSynchronize(stream);   // the ring is reusable once the last piece has left
    for (int i = 0; i < StagingRing::kSlots; ++i)
        if (done[i]) (void)hipEventDestroy(done[i]);
    return rc;
}

// Device -> host through the same pinned ring (results: 128 B per unit).
int d2h_staged(void* dst, const void* src, uint64_t bytes, hipStream_t stream)
{
    if (bytes < (16ull << 20)) {
        if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SVT_OK;
    }
    std::lock_guard<std::mutex> guard(g_ring.lock);
    SVT_TRY(g_ring.ensure());
    const unsigned nt = std::min(host_threads(), 6u);
    // piece k is copied out of its slot while piece k + 1 is on the wire
    uint64_t off = 0, prev_off = 0, prev_len = 0;
    int slot = 0, prev_slot = -1;
    while (off < bytes || prev_slot >= 0) {
        uint64_t len = 0;
        if (off < bytes) {
            len = std::min(StagingRing::kPiece, bytes - off);
            HIP_TRY(hipMemcpyAsync(g_ring.buf[slot], static_cast<const char*>(src) + off, len, hipMemcpyDeviceToHost, stream));
        }
        if (prev_slot >= 0) {
            const char* p0 = static_cast<const char*>(g_ring.buf[prev_slot]);
            char* d0 = static_cast<char*>(dst) + prev_off;
            const uint64_t part = ((prev_len + nt - 1) / nt + 4095) & ~uint64_t(4095);
            parallel_for(nt, [&](uint64_t t) {
                const uint64_t lo = t * part, hi = std::min(prev_len, lo + part);
                if (lo < hi) std::memcpy(d0 + lo, p0 + lo, hi - lo);
            });
        }
        HIP_TRY(hipStreamSynchronize(stream));
        prev_slot = len ? slot : -1;
        prev_off = off;
        prev_len = len;
        off += len;
        slot = (slot + 1) % 2;
    }
    return SVT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// batch object
// ------------------------------------------------------------------------------------------
struct svt_batch {
    int device = 0;
    unsigned flags = 0;
    hipStream_t stream = nullptr;
    uint64_t n_units = 0, n_records = 0, slots = 0;
    uint32_t n_tiles = 0;
    int mode = kSingleLds;
    bool split = true;
    size_t lds_bytes = 0;
    bool have_results = false;
    // device buffers
    uint4* d_tiled = nullptr;
    TileDesc* d_tiles = nullptr;  // dispatch order
    LaneHdr* d_hdr = nullptr;
    double* d_pm = nullptr;
    double* d_l10 = nullptr;
    LibDesc* d_libs = nullptr;
    uint32_t* d_hist = nullptr;
    int32_t* d_thr = nullptr;
    PairWeights* d_wtab = nullptr;
    WgDesc* d_wg = nullptr;
    svt_result* d_out = nullptr;
    KernelArgs args{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

void free_batch(svt_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    auto F = [](void* p) { if (p) (void)hipFree(p); };
    F(b->d_tiled); F(b->d_tiles); F(b->d_hdr); F(b->d_pm); F(b->d_l10); F(b->d_libs);
    F(b->d_hist); F(b->d_thr); F(b->d_wtab); F(b->d_wg); F(b->d_out);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

// device scratch that only lives during svt_batch_create
struct DevScratch {
    void* p = nullptr;
    ~DevScratch() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        HIP_TRY(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        return SVT_OK;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

template <typename T>
int upload(T** dptr, const std::vector<T>& v, hipStream_t s)
{
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(dptr), std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return SVT_OK;
}

template <typename T>
int upload(DevScratch& d, const std::vector<T>& v, hipStream_t s)
{
    SVT_TRY(d.alloc(v.size() * sizeof(T)));
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return SVT_OK;
}

template <bool SSO, bool SPLIT>
const void* kernel_for(int mode)
{
    switch (mode) {
    case kSingleLds: return reinterpret_cast<const void*>(&svt_genotype_kernel<SSO, kSingleLds, SPLIT>);
    case kMultiLds:  return reinterpret_cast<const void*>(&svt_genotype_kernel<SSO, kMultiLds, SPLIT>);
    default:         return reinterpret_cast<const void*>(&svt_genotype_kernel<SSO, kGeneral, SPLIT>);
    }
}

const void* kernel_of(const svt_batch* b)
{
    const bool sso = (b->flags & SVT_FLAG_SSO_ASSOCIATION) != 0;
    if (b->split) return sso ? kernel_for<true, true>(b->mode) : kernel_for<false, true>(b->mode);
    return sso ? kernel_for<true, false>(b->mode) : kernel_for<false, false>(b->mode);
}

int launch_genotype(svt_batch* b)
{
    if (b->n_tiles == 0) return SVT_OK;
    const dim3 grid((b->n_tiles + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlock);
    void* params[] = {&b->args};
    HIP_TRY(hipLaunchKernel(kernel_of(b), grid, block, params, b->lds_bytes, b->stream));
    return SVT_OK;
}

// SVT_TRACE=1 in the environment prints the stage times of svt_batch_create to stderr
struct StageTimer {
    bool on = std::getenv("SVT_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[svt] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// everything of svt_batch_create that needs the device; `b` is freed by the caller on failure
int create_on_device(const svt_evidence_batch* in, svt_batch* b, const uint4* d_records_resident = nullptr)
{
    const uint64_t n = in->n_units;
    const uint64_t n_rec = n ? in->rec_offset[n] : 0;
    StageTimer tm;

    // ---- per-unit record counts + validation of the CSR
    std::vector<uint32_t> nrec(n);
    uint64_t max_f = 0;
    bool wide_var_length = false;
    for (uint64_t u = 0; u < n; ++u) {
        if (in->rec_offset[u + 1] < in->rec_offset[u]) return fail(SVT_ERR_INVALID, "rec_offset not monotone");
        const uint64_t f = in->rec_offset[u + 1] - in->rec_offset[u];
        if (f > 0x3FFFFFFFull) return fail(SVT_ERR_INVALID, "unit with too many records");
        const svt_unit& U = in->units[u];
        if (U.svtype > SVT_SVTYPE_BND) return fail(SVT_ERR_INVALID, "bad svtype");
        if (U.reserved != 0 || (U.flags & ~SVT_UNIT_SKIP)) return fail(SVT_ERR_INVALID, "unit reserved/flags bits must be 0");
        if (U.var_length < -(1 << 30) || U.var_length > (1 << 30)) wide_var_length = true;
        nrec[u] = (uint32_t)f;
        max_f = std::max(max_f, f);
    }

    tm.mark("validate units");
    HostTables T;
    SVT_TRY(build_tables(in, max_f, T));
    if (wide_var_length) T.fast_geometry = false;
    tm.mark("build tables");

    HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&b->ev0));
    HIP_TRY(hipEventCreate(&b->ev1));

    // ---- canonical records to the device; validate them and count the sparse-stream entries
    DevScratch d_off, d_counts, d_err;
    std::unique_lock<std::mutex> csr_guard(g_csr_cache.lock, std::defer_lock);
    const uint4* d_csr = d_records_resident;
    if (!d_csr) {
        csr_guard.lock();   // the cached scratch is ours until we return
        void* d_csr_p = nullptr;
        SVT_TRY(g_csr_cache.acquire(b->device, std::max<uint64_t>(n_rec, 1) * sizeof(uint4), &d_csr_p));
        d_csr = static_cast<const uint4*>(d_csr_p);
        tm.mark("stream/event/alloc");
        SVT_TRY(h2d_staged(d_csr_p, in->records, n_rec * sizeof(uint4), b->stream));
        tm.mark("H2D records (staged)");
    }
    SVT_TRY(d_off.alloc((n + 1) * sizeof(uint64_t)));
    if (n) HIP_TRY(hipMemcpyAsync(d_off.p, in->rec_offset, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, b->stream));
    SVT_TRY(d_counts.alloc(n * sizeof(uint4)));
    SVT_TRY(d_err.alloc(sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), b->stream));
    std::vector<uint4> counts(n);
    uint32_t err_bits = 0;
    if (n) {
        hipLaunchKernelGGL(svt_scan_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, b->stream,
                           d_csr, d_off.as<uint64_t>(), n, in->n_libs, d_counts.as<uint4>(),
                           d_err.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(counts.data(), d_counts.p, n * sizeof(uint4), hipMemcpyDeviceToHost, b->stream));
    }
    HIP_TRY(hipMemcpyAsync(&err_bits, d_err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    tm.mark("scan kernel + counts D2H");
    if (err_bits) {
        std::string m = "invalid evidence records:";
        if (err_bits & 2u) m += " straddle bits without HAS_PAIR;";
        if (err_bits & 4u) m += " lib index >= n_libs;";
        if (err_bits & 8u) m += " reserved/undefined bits set;";
        if (err_bits & 16u) m += " negative ospan_len;";
        return fail(SVT_ERR_INVALID, m);
    }

    // ---- tiling
    std::vector<uint32_t> len_a(n), len_b(n);
    for (uint64_t u = 0; u < n; ++u) {
        if (b->split) {
            len_a[u] = (counts[u].x + 1) / 2;   // two 8-byte entries per 16-byte row slot
            len_b[u] = (counts[u].y + 1) / 2;
        } else {
            len_a[u] = nrec[u];
            len_b[u] = 0;
        }
    }
    Tiling G;
    build_tiling(in, nrec, len_a, len_b, counts, G);
    if (G.tiles.size() > 0xFFFFFFF0ull / kWave) return fail(SVT_ERR_INVALID, "too many tiles");
    b->n_tiles = (uint32_t)G.tiles.size();
    b->slots = G.slots;
    // dispatch order: tiles stay in groups of 4 consecutive (= one workgroup, similar length, same
    // libraries); the groups go longest first (LPT) so the tail of the grid is made of short tiles
    const uint32_t n_groups = (b->n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    std::vector<uint32_t> group_order(n_groups);
    std::vector<uint64_t> group_cost(n_groups, 0);
    for (uint32_t g = 0; g < n_groups; ++g) {
        group_order[g] = g;
        for (uint32_t t = g * kWavesPerBlock; t < std::min(b->n_tiles, (g + 1) * kWavesPerBlock); ++t)
            group_cost[g] += G.tiles[t].rows_a + G.tiles[t].rows_b;
    }
    std::stable_sort(group_order.begin(), group_order.end(),
                     [&](uint32_t x, uint32_t y) { return group_cost[x] > group_cost[y]; });
    std::vector<TileDesc> dispatch;
    std::vector<WgDesc> windows;
    dispatch.reserve((size_t)n_groups * kWavesPerBlock);
    uint32_t max_win_libs = 1, max_win_bins = 1;
    for (uint32_t g : group_order) {
        uint32_t lo = 0xffffffffu, hi = 0;
        for (uint32_t k = 0; k < (uint32_t)kWavesPerBlock; ++k) {
            const uint32_t t = g * kWavesPerBlock + k;
            if (t < b->n_tiles) {
                dispatch.push_back(G.tiles[t]);
                lo = std::min(lo, G.tile_lib_lo[t]);
                hi = std::max(hi, G.tile_lib_hi[t]);
            } else {
                TileDesc pad{};
                pad.lane_base = kPadUnit;          // marks a tile that does not exist
                dispatch.push_back(pad);
            }
        }
        WgDesc w{};
        w.lib_lo = lo;
        w.lib_cnt = hi - lo + 1;
        w.bin_lo = T.libs[lo].tab_off;
        w.bin_cnt = T.libs[hi].tab_off + T.libs[hi].n_bins + 1 - w.bin_lo;
        windows.push_back(w);
        max_win_libs = std::max(max_win_libs, w.lib_cnt);
        max_win_bins = std::max(max_win_bins, w.bin_cnt);
    }

    tm.mark("tiling (host sort)");
    // ---- resident device objects
    DevScratch d_tiles_store, d_lane_src, d_lane_nrec;
    SVT_TRY(upload(&b->d_tiles, dispatch, b->stream));
    SVT_TRY(upload(d_tiles_store, G.tiles, b->stream));
    SVT_TRY(upload(&b->d_hdr, G.hdr, b->stream));
    SVT_TRY(upload(d_lane_src, G.lane_src, b->stream));
    SVT_TRY(upload(d_lane_nrec, G.lane_nrec, b->stream));
    SVT_TRY(upload(&b->d_pm, T.pm, b->stream));
    SVT_TRY(upload(&b->d_l10, T.l10, b->stream));
    SVT_TRY(upload(&b->d_libs, T.libs, b->stream));
    SVT_TRY(upload(&b->d_hist, T.hist, b->stream));
    SVT_TRY(upload(&b->d_thr, T.thr, b->stream));
    SVT_TRY(upload(&b->d_wtab, T.wtab, b->stream));
    SVT_TRY(upload(&b->d_wg, windows, b->stream));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_tiled), (G.slots + kTailPadRows * kWave) * sizeof(uint4)));
    HIP_TRY(hipMemsetAsync(b->d_tiled + G.slots, 0, kTailPadRows * kWave * sizeof(uint4), b->stream));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_out), std::max<uint64_t>(n, 1) * sizeof(svt_result)));

    // ---- re-tile on the device
    if (b->n_tiles) {
        RepackArgs ra{};
        ra.csr = d_csr;
        ra.lane_src = d_lane_src.as<uint64_t>();
        ra.lane_nrec = d_lane_nrec.as<uint32_t>();
        ra.tiles = d_tiles_store.as<TileDesc>();
        ra.tiled = b->d_tiled;
        ra.n_tiles = b->n_tiles;
        const dim3 grid((b->n_tiles + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlock);
        if (b->split) hipLaunchKernelGGL(svt_repack_split_kernel, grid, block, 0, b->stream, ra);
        else hipLaunchKernelGGL(svt_repack_dense_kernel, grid, block, 0, b->stream, ra);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(b->stream));  // scratch buffers are released on return
    tm.mark("uploads + repack kernel");

    // ---- kernel arguments
    KernelArgs& a = b->args;
    a.tiled = b->d_tiled;
    a.tiles = b->d_tiles;
    a.hdr = b->d_hdr;
    a.pm = b->d_pm;
    a.l10 = b->d_l10;
    a.libs = b->d_libs;
    a.hist = b->d_hist;
    a.thr = b->d_thr;
    a.wtab = b->d_wtab;
    a.n_l10 = (uint32_t)T.l10.size();
    a.n_libs = in->n_libs;
    a.total_bins = (uint32_t)T.hist.size();
    a.n_tiles = n_groups * kWavesPerBlock;   // the dispatch list is padded to whole workgroups
    a.l10_in_lds = a.n_l10 <= kMaxL10Lds ? 1u : 0u;
    a.n_units = n;
    a.out = b->d_out;
    a.lib0 = T.libs[0];
    fill_gt_consts(a.c, in->split_weight, in->disc_weight);

    a.wg = b->d_wg;
    // kernel flavour: tables in LDS when the 32-bit geometry holds and the largest per-workgroup
    // library window fits the LDS budget; otherwise the general kernel reads them through L2
    if (T.fast_geometry && (size_t)max_win_bins * 8 <= kMaxLdsTableBytes) {
        b->mode = in->n_libs == 1 ? kSingleLds : kMultiLds;
        a.lds_libs = in->n_libs == 1 ? 1u : max_win_libs;
        a.lds_bins = in->n_libs == 1 ? a.total_bins : max_win_bins;
    } else {
        b->mode = kGeneral;
        a.lds_libs = in->n_libs;   // descriptors only
        a.lds_bins = 0;
    }
    const uint32_t n_l10_lds = a.l10_in_lds ? ((a.n_l10 + 1u) & ~1u) : 0u;
    b->lds_bytes = 256 * 8 + 32 * sizeof(PairWeights) + (size_t)n_l10_lds * 8 +
                   (size_t)a.lds_libs * sizeof(LibDesc) + (size_t)a.lds_bins * 8;
    b->lds_bytes = (b->lds_bytes + 15) & ~size_t(15);
    if (b->lds_bytes > 160 * 1024) return fail(SVT_ERR_INVALID, "LDS budget exceeded");
    if (b->lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(kernel_of(b), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes));
    return SVT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int svt_version(void) { return SVT_ABI_VERSION; }

int svt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* svt_last_error(void) { return g_err.c_str(); }

int svt_batch_create(const svt_evidence_batch* in, int device, unsigned flags, svt_batch** out)
{
    if (!in || !out) return fail(SVT_ERR_INVALID, "null argument");
    *out = nullptr;
    const uint64_t n = in->n_units;
    if (flags & ~(SVT_FLAG_SSO_ASSOCIATION | SVT_FLAG_DENSE_LAYOUT)) return fail(SVT_ERR_INVALID, "unknown flag bits");
    if (n >= 0xFFFFFFF0ull) return fail(SVT_ERR_INVALID, "too many units in one batch (< 2^32)");
    if (in->n_libs == 0 || in->n_libs > 256 || !in->libs) return fail(SVT_ERR_INVALID, "n_libs must be 1..256");
    if (n && (!in->rec_offset || !in->units)) return fail(SVT_ERR_INVALID, "null unit arrays");
    if (n && in->rec_offset[0] != 0) return fail(SVT_ERR_INVALID, "rec_offset[0] must be 0");
    if (n && in->rec_offset[n] && !in->records) return fail(SVT_ERR_INVALID, "null records");
    if (!(in->split_weight >= 0.0) || !(in->disc_weight >= 0.0) || !std::isfinite(in->split_weight) ||
        !std::isfinite(in->disc_weight))
        return fail(SVT_ERR_INVALID, "weights must be finite and >= 0");

    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    svt_batch* b = new (std::nothrow) svt_batch();
    if (!b) return fail(SVT_ERR_NOMEM, "out of host memory");
    b->device = device;
    b->flags = flags;
    b->split = (flags & SVT_FLAG_DENSE_LAYOUT) == 0;
    b->n_units = n;
    b->n_records = n ? in->rec_offset[n] : 0;
    const int rc = create_on_device(in, b);
    if (rc != SVT_OK) {
        const std::string keep = g_err;
        free_batch(b);
        g_err = keep;
        return rc;
    }
    *out = b;
    return SVT_OK;
}

int svt_batch_create_from_fragments(const svt_fragment_batch* in, int device, unsigned flags,
                                    svt_record* records_out, svt_batch** out)
{
    if (!in || !out) return fail(SVT_ERR_INVALID, "null argument");
    *out = nullptr;
    const uint64_t n = in->n_units;
    if (flags & ~(SVT_FLAG_SSO_ASSOCIATION | SVT_FLAG_DENSE_LAYOUT)) return fail(SVT_ERR_INVALID, "unknown flag bits");
    if (n >= 0xFFFFFFF0ull) return fail(SVT_ERR_INVALID, "too many units in one batch (< 2^32)");
    if (in->n_libs == 0 || in->n_libs > 256 || !in->libs) return fail(SVT_ERR_INVALID, "n_libs must be 1..256");
    if (n && (!in->frag_offset || !in->breakpoints)) return fail(SVT_ERR_INVALID, "null unit arrays");
    if (n && in->frag_offset[0] != 0) return fail(SVT_ERR_INVALID, "frag_offset[0] must be 0");
    const uint64_t n_frag = n ? in->frag_offset[n] : 0;
    if (n_frag && !in->fragments) return fail(SVT_ERR_INVALID, "null fragments");
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    // unit headers and the fragment -> unit map
    std::vector<svt_unit> units(n);
    std::vector<uint32_t> frag_unit(n_frag);
    for (uint64_t u = 0; u < n; ++u) {
        const svt_breakpoint& bp = in->breakpoints[u];
        if (in->frag_offset[u + 1] < in->frag_offset[u]) return fail(SVT_ERR_INVALID, "frag_offset not monotone");
        if (bp.svtype > SVT_SVTYPE_BND) return fail(SVT_ERR_INVALID, "bad svtype");
        svt_unit U{};
        U.var_length = bp.svtype == SVT_SVTYPE_DEL ? bp.var_length : 0;
        const int64_t delta = (int64_t)bp.pos_b - (int64_t)bp.pos_a;            // classic.py:339
        U.pos_delta = (int32_t)std::max<int64_t>(INT32_MIN, std::min<int64_t>(INT32_MAX, delta));
        U.sample = bp.sample;
        U.svtype = bp.svtype;
        U.flags = (bp.flags & SVT_BP_SKIP) ? SVT_UNIT_SKIP : 0;
        units[u] = U;
        for (uint64_t j = in->frag_offset[u]; j < in->frag_offset[u + 1]; ++j) frag_unit[j] = (uint32_t)u;
    }
    // library descriptors (the flank of is_pair_straddle is lib.mean + lib.sd * 3)
    std::vector<LibDesc> libs(in->n_libs);
    for (uint32_t l = 0; l < in->n_libs; ++l) {
        if (!std::isfinite(in->libs[l].mean) || !std::isfinite(in->libs[l].sd)) return fail(SVT_ERR_INVALID, "library moments not finite");
        libs[l].v_nondel = in->libs[l].mean + in->libs[l].sd * 3;
    }

    // geometry on the device
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{s};
    StageTimer tm;
    DevScratch d_frags, d_frag_unit, d_bps, d_libs, d_records, d_err;
    SVT_TRY(d_frags.alloc(n_frag * sizeof(svt_fragment)));
    SVT_TRY(h2d_staged(d_frags.p, in->fragments, n_frag * sizeof(svt_fragment), s));
    tm.mark("H2D fragment summaries");
    SVT_TRY(upload(d_frag_unit, frag_unit, s));
    SVT_TRY(d_bps.alloc(n * sizeof(svt_breakpoint)));
    if (n) HIP_TRY(hipMemcpyAsync(d_bps.p, in->breakpoints, n * sizeof(svt_breakpoint), hipMemcpyHostToDevice, s));
    SVT_TRY(upload(d_libs, libs, s));
    SVT_TRY(d_records.alloc(n_frag * sizeof(uint4)));
    SVT_TRY(d_err.alloc(sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), s));
    if (n_frag) {
        GeomArgs g{};
        g.frags = d_frags.as<uint4>();
        g.frag_unit = d_frag_unit.as<uint32_t>();
        g.bps = d_bps.as<svt_breakpoint>();
        g.libs = d_libs.as<LibDesc>();
        g.n_frags = n_frag;
        g.n_libs = in->n_libs;
        g.min_aligned = in->min_aligned;
        g.split_slop = in->split_slop;
        g.records = d_records.as<uint4>();
        g.err = d_err.as<uint32_t>();
        hipLaunchKernelGGL(svt_geometry_kernel, dim3((unsigned)((n_frag + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, g);
        HIP_TRY(hipGetLastError());
    }
    uint32_t err_bits = 0;
    HIP_TRY(hipMemcpyAsync(&err_bits, d_err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (records_out && n_frag)
        HIP_TRY(hipMemcpyAsync(records_out, d_records.p, n_frag * sizeof(uint4), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    tm.mark("geometry kernel (+ copies)");
    if (err_bits) return fail(SVT_ERR_INVALID, "invalid fragment summaries: library index >= n_libs");

    // the resident batch, from the records that are already in HBM
    svt_evidence_batch eb{};
    eb.n_units = n;
    eb.rec_offset = in->frag_offset;
    eb.units = units.data();
    eb.records = nullptr;
    eb.n_libs = in->n_libs;
    eb.libs = in->libs;
    eb.split_weight = in->split_weight;
    eb.disc_weight = in->disc_weight;
    if (!(eb.split_weight >= 0.0) || !(eb.disc_weight >= 0.0) || !std::isfinite(eb.split_weight) ||
        !std::isfinite(eb.disc_weight))
        return fail(SVT_ERR_INVALID, "weights must be finite and >= 0");
    svt_batch* b = new (std::nothrow) svt_batch();
    if (!b) return fail(SVT_ERR_NOMEM, "out of host memory");
    b->device = device;
    b->flags = flags;
    b->split = (flags & SVT_FLAG_DENSE_LAYOUT) == 0;
    b->n_units = n;
    b->n_records = n_frag;
    const int rc = create_on_device(&eb, b, d_records.as<uint4>());
    if (rc != SVT_OK) {
        const std::string keep = g_err;
        free_batch(b);
        g_err = keep;
        return rc;
    }
    *out = b;
    return SVT_OK;
}

int svt_batch_genotype(svt_batch* b, int sync)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    HIP_TRY(hipSetDevice(b->device));
    SVT_TRY(launch_genotype(b));
    b->have_results = true;
    if (sync) HIP_TRY(hipStreamSynchronize(b->stream));
    return SVT_OK;
}

int svt_batch_genotype_n(svt_batch* b, int iters)
{
    if (!b || iters <= 0) return fail(SVT_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    for (int i = 0; i < iters; ++i) SVT_TRY(launch_genotype(b));
    b->have_results = true;
    return SVT_OK;
}

int svt_batch_genotype_timed(svt_batch* b, int iters, float* ms_total)
{
    if (!b || !ms_total || iters <= 0) return fail(SVT_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipEventRecord(b->ev0, b->stream));
    for (int i = 0; i < iters; ++i) SVT_TRY(launch_genotype(b));
    HIP_TRY(hipEventRecord(b->ev1, b->stream));
    HIP_TRY(hipEventSynchronize(b->ev1));
    HIP_TRY(hipEventElapsedTime(ms_total, b->ev0, b->ev1));
    b->have_results = true;
    return SVT_OK;
}

int svt_batch_results(svt_batch* b, svt_result* out, uint64_t n_units)
{
    if (!b || (!out && n_units)) return fail(SVT_ERR_INVALID, "null argument");
    if (!b->have_results) return fail(SVT_ERR_STATE, "svt_batch_genotype has not run");
    if (n_units != b->n_units) return fail(SVT_ERR_INVALID, "results n_units mismatch");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));   // the pass that produced the records
    return d2h_staged(out, b->args.out, b->n_units * sizeof(svt_result), b->stream);
}

int svt_batch_device_results(svt_batch* b, svt_result** dev)
{
    if (!b || !dev) return fail(SVT_ERR_INVALID, "null argument");
    *dev = b->args.out;
    return SVT_OK;
}

int svt_batch_bind_device_results(svt_batch* b, svt_result* dev)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    if (dev && (reinterpret_cast<uintptr_t>(dev) & 127u)) return fail(SVT_ERR_INVALID, "result buffer must be 128-byte aligned");
    b->args.out = dev ? dev : b->d_out;
    b->have_results = false;
    return SVT_OK;
}

int svt_batch_bytes(const svt_batch* b, uint64_t* algorithmic, uint64_t* resident)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    if (algorithmic) *algorithmic = 16 * b->n_records + (16 + 96) * b->n_units;
    if (resident) *resident = 16 * b->slots + (uint64_t)b->n_tiles * kWave * sizeof(LaneHdr);
    return SVT_OK;
}

int svt_bayes_gt(const int32_t* ref, const int32_t* alt, const uint8_t* is_dup, uint64_t n, double* out,
                 int device)
{
    if (n == 0) return SVT_OK;
    if (!ref || !alt || !is_dup || !out) return fail(SVT_ERR_INVALID, "null argument");
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    int64_t max_total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (ref[i] < 0 || alt[i] < 0) return fail(SVT_ERR_INVALID, "negative read count");
        max_total = std::max<int64_t>(max_total, (int64_t)ref[i] + alt[i]);
    }
    if (max_total >= (1 << 24)) return fail(SVT_ERR_INVALID, "ref + alt must be < 2^24");
    std::vector<double> l10((size_t)max_total + 2);
    l10[0] = 0.0;
    for (size_t i = 1; i < l10.size(); ++i) l10[i] = py_log10((double)i);
    GtConsts c{};
    fill_gt_consts(c, 1.0, 1.0);
    HIP_TRY(hipSetDevice(device));
    DevScratch d_ref, d_alt, d_dup, d_l10, d_out;
    SVT_TRY(d_ref.alloc(n * sizeof(int32_t)));
    SVT_TRY(d_alt.alloc(n * sizeof(int32_t)));
    SVT_TRY(d_dup.alloc(n));
    SVT_TRY(d_l10.alloc(l10.size() * sizeof(double)));
    SVT_TRY(d_out.alloc(n * 4 * sizeof(double)));
    HIP_TRY(hipMemcpy(d_ref.p, ref, n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_alt.p, alt, n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_dup.p, is_dup, n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_l10.p, l10.data(), l10.size() * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(svt_bayes_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                       d_ref.as<int32_t>(), d_alt.as<int32_t>(), d_dup.as<uint8_t>(), n, d_l10.as<double>(), c,
                       d_out.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, n * 4 * sizeof(double), hipMemcpyDeviceToHost));
    return SVT_OK;
}

void* svt_batch_stream(svt_batch* b) { return b ? (void*)b->stream : nullptr; }

void svt_batch_destroy(svt_batch* b) { free_batch(b); }

void svt_trim(void) { g_csr_cache.trim(); }

int svt_genotype(const svt_evidence_batch* in, svt_result* out, int device, unsigned flags)
{
    svt_batch* b = nullptr;
    SVT_TRY(svt_batch_create(in, device, flags, &b));
    int rc = svt_batch_genotype(b, 1);
    if (rc == SVT_OK) rc = svt_batch_results(b, out, in->n_units);
    const std::string keep = g_err;
    svt_batch_destroy(b);
    g_err = keep;
    return rc;
}

}  // extern "C"
