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
//     the j-th record of its 64 units back to back, so every wave-level load is one contiguous
//     1 KiB global_load_dwordx4.  Units are sorted by record count inside 4096-unit chunks so
//     that the zero-padding of a tile stays ~1 %.
//   * all look-up tables (prob_mapq, insert-size histogram + p_concordant thresholds, log10)
//     are built on the host with the same libm CPython uses and staged in LDS per workgroup.
//   * HBM-bound byte/integer/fp64 streaming: no MFMA anywhere (nothing here is a contraction).
//
// There is no CPU fallback in this file.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/svtyper_hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------
constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kWavesPerBlock = 4;    // 256-thread workgroups
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr uint32_t kChunkUnits = 4096;  // sort window (units); results scatter stays inside it
constexpr uint32_t kPadUnit = 0xFFFFFFFFu;
constexpr uint32_t kMaxLdsTableBytes = 96 * 1024;  // hist+T budget before falling back to global
constexpr uint32_t kMaxL10Lds = 4096;              // log10 table entries kept in LDS (32 KiB)

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

// ------------------------------------------------------------------------------------------
// device-side structures
// ------------------------------------------------------------------------------------------
struct LibDesc {          // 32 B, one per library
    uint32_t tab_off;     // offset of this library's bins inside hist[] / thr[]
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

struct TileDesc {         // 16 B, one per 64-unit tile, stored in dispatch (longest-first) order
    uint64_t rec_base;    // first record of the tile inside tiled[]
    uint32_t rows;        // records per lane (max F of the tile)
    uint32_t lane_base;   // first LaneHdr of the tile
};

struct GtConsts {
    double lgp[2][3];     // [is_dup][genotype] log(p)/log(10)      (statistics.py:33-35)
    double lg1p[2][3];    // [is_dup][genotype] log(1-p)/log(10)
    double ln10;          // log(10.0)
    double x_uflow;       // smallest x with libm pow(10.0, x) > 0
    double split_weight;
    double disc_weight;
};

struct KernelArgs {
    const uint4* tiled;
    const TileDesc* tiles;
    const LaneHdr* hdr;
    const double* pm;        // 256
    const double* l10;       // n_l10
    const LibDesc* libs;     // n_libs
    const uint32_t* hist;    // total_bins
    const int32_t* thr;      // total_bins
    uint32_t n_l10;
    uint32_t n_libs;
    uint32_t total_bins;
    uint32_t n_tiles;
    uint32_t l10_in_lds;
    uint32_t pad0;
    uint64_t n_units;
    double* gl;              // [3][n]
    double* sq;              // [n]
    double* tallies;         // [5][n]
    int32_t* counts;         // [11][n]
    int8_t* gt;              // [n]
    GtConsts c;
};

// ------------------------------------------------------------------------------------------
// genotype kernel
// ------------------------------------------------------------------------------------------
struct Tables {
    const double* pm;        // LDS
    const LibDesc* libs;     // LDS
    const uint32_t* hist;    // LDS or global
    const int32_t* thr;      // LDS or global
};

struct Acc {
    double ref_seq, alt_seq, alt_clip, ref_span, alt_span;
    double l_ref_seq, l_alt_seq, l_alt_clip;  // sso fragment-local sums
};

// One evidence record.  All adds are unconditional adds of (cond ? x : +0.0): x + 0.0 == x
// bit-for-bit for the non-negative sums involved, so predication never changes a result.
template <bool SSO>
__device__ __forceinline__ void tally_record(const uint4 w, const Tables& t, const bool is_del,
                                             const int32_t var_length, const double pos_delta_d,
                                             Acc& a)
{
    const uint32_t f = w.w;
    const double pm_a = t.pm[w.y & 0xffu];
    const double pm_b = t.pm[(w.y >> 8) & 0xffu];
    const double ps0l = t.pm[(w.y >> 16) & 0xffu];
    const double ps0r = t.pm[w.y >> 24];
    const double ps1l = t.pm[w.z & 0xffu];
    const double ps1r = t.pm[(w.z >> 8) & 0xffu];
    const LibDesc lib = t.libs[(w.z >> 16) & 0xffu];

    // ---- reference split-read evidence (classic.py:306-311)
    const double rsa = (f & SVT_REC_REFSEQ_A) ? pm_a : 0.0;
    const double rsb = (f & SVT_REC_REFSEQ_B) ? pm_b : 0.0;
    // ---- alternate split-read evidence (classic.py:317-328):
    //      p_alt = (pm(left) * L + pm(right) * R) / 2.0
    const double p0 = (((f & SVT_REC_S0_L) ? ps0l : 0.0) + ((f & SVT_REC_S0_R) ? ps0r : 0.0)) * 0.5;
    const double p1 = (((f & SVT_REC_S1_L) ? ps1l : 0.0) + ((f & SVT_REC_S1_R) ? ps1r : 0.0)) * 0.5;
    const double as0 = (f & SVT_REC_S0_SOFT) ? 0.0 : p0;
    const double ac0 = (f & SVT_REC_S0_SOFT) ? p0 : 0.0;
    const double as1 = (f & SVT_REC_S1_SOFT) ? 0.0 : p1;
    const double ac1 = (f & SVT_REC_S1_SOFT) ? p1 : 0.0;

    if (SSO) {
        // singlesample.py:246-276,367-372: per-fragment sums starting from 0, flushed into the
        // site totals when the next fragment starts
        const bool cont = (f & SVT_REC_CONTINUATION) != 0;
        a.ref_seq += cont ? 0.0 : a.l_ref_seq;
        a.alt_seq += cont ? 0.0 : a.l_alt_seq;
        a.alt_clip += cont ? 0.0 : a.l_alt_clip;
        a.l_ref_seq = ((cont ? a.l_ref_seq : 0.0) + rsa) + rsb;
        a.l_alt_seq = ((cont ? a.l_alt_seq : 0.0) + as0) + as1;
        a.l_alt_clip = ((cont ? a.l_alt_clip : 0.0) + ac0) + ac1;
    } else {
        a.ref_seq = (a.ref_seq + rsa) + rsb;
        a.alt_seq = (a.alt_seq + as0) + as1;
        a.alt_clip = (a.alt_clip + ac0) + ac1;
    }

    // ---- paired-end evidence (classic.py:339-408)
    const bool small_del = is_del && (pos_delta_d < lib.sd2);           // :339, :383
    const bool alt_st = !small_del && (f & SVT_REC_ALT_STRADDLE);
    const bool rs_a = !small_del && (f & SVT_REC_REF_STRADDLE_A);
    const bool rs_b = !small_del && (f & SVT_REC_REF_STRADDLE_B);
    const bool both = rs_a && rs_b;
    const bool need_ref = (rs_a || rs_b) && (!both || is_del);          // :398-401

    // p_concordant (parsers.py:861-882) as an integer test: with d1 = hist[o]/N fixed, the
    // reference's binary64 expression d1*0.95/(0.95*d1 + 0.05*d2) > 0.5 is monotone in
    // h2 = hist[o - v]; thr[o] is the largest h2 for which it still holds (evaluated on the host
    // with the reference's own expression), -1 where hist[o] == 0 (p == 0 or ZeroDivisionError).
    const int32_t o = (int32_t)w.x;
    const int64_t i1 = (int64_t)o - (int64_t)lib.key_min;
    const bool in1 = (uint64_t)i1 < (uint64_t)lib.n_bins;
    const int32_t thr1 = in1 ? t.thr[lib.tab_off + (uint32_t)(in1 ? i1 : 0)] : -1;
    int64_t key2;
    bool key2_ok = true;
    if (is_del) {
        key2 = (int64_t)o - (int64_t)var_length;
    } else {
        // var_length is None: Counter key is the FLOAT o - (mean + 3 sd); it only matches an
        // integer key when it is integral (parsers.py:874-878)
        const double kf = (double)o - lib.v_nondel;
        key2_ok = (kf == floor(kf)) && (fabs(kf) < 4.0e9);
        key2 = key2_ok ? (int64_t)kf : 0;
    }
    const int64_t i2 = key2 - (int64_t)lib.key_min;
    const bool in2 = key2_ok && ((uint64_t)i2 < (uint64_t)lib.n_bins);
    const int32_t h2 = in2 ? (int32_t)t.hist[lib.tab_off + (uint32_t)(in2 ? i2 : 0)] : 0;
    const bool p_conc = h2 <= thr1;

    const double pp = pm_a * pm_b;
    // DEL: alt_span += (1 - p_conc) * pmA * pmB (:363-364); others: pmA * pmB (:376-377)
    a.alt_span += (alt_st && !(is_del && p_conc)) ? pp : 0.0;
    // ref_span += (A + B) * (p_conc * pmA * pmB) / 2 (:404-405): 2*pp/2 == pp, 1*pp/2 == pp*0.5
    a.ref_span += (need_ref && p_conc) ? (both ? pp : pp * 0.5) : 0.0;
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

template <bool SSO, bool LDS_TABLES>
__global__ __launch_bounds__(kBlock) void svt_genotype_kernel(const KernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* s_pm = reinterpret_cast<double*>(smem);
    const uint32_t n_l10_lds = a.l10_in_lds ? ((a.n_l10 + 1u) & ~1u) : 0u;
    double* s_l10 = s_pm + 256;
    LibDesc* s_lib = reinterpret_cast<LibDesc*>(s_l10 + n_l10_lds);
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_lib + a.n_libs);
    int32_t* s_thr = reinterpret_cast<int32_t*>(s_hist + (LDS_TABLES ? a.total_bins : 0u));

    // ---- stage the tables in LDS (L2-resident after the first workgroups)
    for (uint32_t i = threadIdx.x; i < 256; i += kBlock) s_pm[i] = a.pm[i];
    if (a.l10_in_lds)
        for (uint32_t i = threadIdx.x; i < a.n_l10; i += kBlock) s_l10[i] = a.l10[i];
    for (uint32_t i = threadIdx.x; i < a.n_libs * (sizeof(LibDesc) / 8); i += kBlock)
        reinterpret_cast<uint64_t*>(s_lib)[i] = reinterpret_cast<const uint64_t*>(a.libs)[i];
    if (LDS_TABLES) {
        for (uint32_t i = threadIdx.x; i < a.total_bins; i += kBlock) {
            s_hist[i] = a.hist[i];
            s_thr[i] = a.thr[i];
        }
    }
    __syncthreads();

    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x % kWave;
    const uint32_t tile_idx = blockIdx.x * kWavesPerBlock + wave;
    if (tile_idx >= a.n_tiles) return;

    const TileDesc td = a.tiles[tile_idx];
    const LaneHdr h = a.hdr[td.lane_base + lane];
    const uint32_t svtype = h.packed & 0xffu;
    const uint32_t uflags = (h.packed >> 8) & 0xffu;
    const bool is_del = svtype == SVT_SVTYPE_DEL;
    const double pos_delta_d = (double)h.pos_delta;

    Tables t;
    t.pm = s_pm;
    t.libs = s_lib;
    t.hist = LDS_TABLES ? s_hist : a.hist;
    t.thr = LDS_TABLES ? s_thr : a.thr;

    Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

    // ---- stream the tile: row j is one contiguous 1 KiB line for the wave
    const uint4* __restrict__ p = a.tiled + td.rec_base + lane;
    const uint32_t rows = td.rows;
    uint32_t j = 0;
    for (; j + 4 <= rows; j += 4) {
        const uint4 w0 = p[(uint64_t)(j + 0) * kWave];
        const uint4 w1 = p[(uint64_t)(j + 1) * kWave];
        const uint4 w2 = p[(uint64_t)(j + 2) * kWave];
        const uint4 w3 = p[(uint64_t)(j + 3) * kWave];
        tally_record<SSO>(w0, t, is_del, h.var_length, pos_delta_d, acc);
        tally_record<SSO>(w1, t, is_del, h.var_length, pos_delta_d, acc);
        tally_record<SSO>(w2, t, is_del, h.var_length, pos_delta_d, acc);
        tally_record<SSO>(w3, t, is_del, h.var_length, pos_delta_d, acc);
    }
    for (; j < rows; ++j) {
        const uint4 w0 = p[(uint64_t)j * kWave];
        tally_record<SSO>(w0, t, is_del, h.var_length, pos_delta_d, acc);
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

    // ---- scatter to the unit's slot (stays inside the unit's 4096-unit chunk)
    const uint64_t u = h.unit;
    const uint64_t n = a.n_units;
    a.gl[u] = gl[0];
    a.gl[n + u] = gl[1];
    a.gl[2 * n + u] = gl[2];
    a.sq[u] = sq;
    a.tallies[0 * n + u] = ref_seq;
    a.tallies[1 * n + u] = alt_seq;
    a.tallies[2 * n + u] = alt_clip;
    a.tallies[3 * n + u] = ref_span;
    a.tallies[4 * n + u] = alt_span;
#pragma unroll
    for (int i = 0; i < SVT_N_COUNTS; ++i) a.counts[(uint64_t)i * n + u] = cnt[i];
    a.gt[u] = (int8_t)gt;
}

// ------------------------------------------------------------------------------------------
// re-tiling kernel: CSR records -> lane-interleaved tiles (runs once per batch)
// ------------------------------------------------------------------------------------------
struct RepackArgs {
    const uint4* csr;
    const uint64_t* lane_src;   // per tile lane: first CSR record of the unit
    const uint32_t* lane_nrec;  // per tile lane: F (0 for padding lanes)
    const TileDesc* tiles;      // in storage order
    uint4* tiled;
    uint32_t n_tiles;
    uint32_t n_libs;
    uint32_t* err;              // OR of violation bits
};

__global__ __launch_bounds__(kBlock) void svt_repack_kernel(const RepackArgs a)
{
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x % kWave;
    const uint32_t tile_idx = blockIdx.x * kWavesPerBlock + wave;
    if (tile_idx >= a.n_tiles) return;
    const TileDesc td = a.tiles[tile_idx];
    const uint64_t src = a.lane_src[td.lane_base + lane];
    const uint32_t nrec = a.lane_nrec[td.lane_base + lane];
    uint32_t bad = 0;
    for (uint32_t j = 0; j < td.rows; ++j) {
        uint4 w = make_uint4(0, 0, 0, 0);
        if (j < nrec) {
            w = a.csr[src + j];
            const uint32_t f = w.w;
            // contract of include/svtyper_hip.h: split bits need PRESENT, straddle bits need
            // HAS_PAIR, lib < n_libs, reserved == 0, no undefined flag bits
            if (!(f & SVT_REC_S0_PRESENT) && (f & (SVT_REC_S0_SOFT | SVT_REC_S0_L | SVT_REC_S0_R))) bad |= 1u;
            if (!(f & SVT_REC_S1_PRESENT) && (f & (SVT_REC_S1_SOFT | SVT_REC_S1_L | SVT_REC_S1_R))) bad |= 1u;
            if (!(f & SVT_REC_HAS_PAIR) &&
                (f & (SVT_REC_ALT_STRADDLE | SVT_REC_REF_STRADDLE_A | SVT_REC_REF_STRADDLE_B))) bad |= 2u;
            if (((w.z >> 16) & 0xffu) >= a.n_libs) bad |= 4u;
            if ((w.z >> 24) != 0u || (f >> 15) != 0u) bad |= 8u;
            if ((int32_t)w.x < 0) bad |= 16u;
        }
        a.tiled[td.rec_base + (uint64_t)j * kWave + lane] = w;
    }
    if (bad) atomicOr(a.err, bad);
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

}  // namespace

// ------------------------------------------------------------------------------------------
// batch object
// ------------------------------------------------------------------------------------------
struct svt_batch {
    int device = 0;
    unsigned flags = 0;
    hipStream_t stream = nullptr;
    uint64_t n_units = 0, n_records = 0, tiled_records = 0;
    uint32_t n_tiles = 0;
    bool lds_tables = true;
    size_t lds_bytes = 0;
    bool have_results = false;
    bool bound_external = false;
    // device buffers
    uint4* d_tiled = nullptr;
    TileDesc* d_tiles = nullptr;  // dispatch order
    LaneHdr* d_hdr = nullptr;
    double* d_pm = nullptr;
    double* d_l10 = nullptr;
    LibDesc* d_libs = nullptr;
    uint32_t* d_hist = nullptr;
    int32_t* d_thr = nullptr;
    double* d_gl = nullptr;
    double* d_sq = nullptr;
    double* d_tallies = nullptr;
    int32_t* d_counts = nullptr;
    int8_t* d_gt = nullptr;
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
    F(b->d_hist); F(b->d_thr); F(b->d_gl); F(b->d_sq); F(b->d_tallies); F(b->d_counts); F(b->d_gt);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

template <typename T>
int upload(T** dptr, const std::vector<T>& v, hipStream_t s)
{
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(dptr), bytes));
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return SVT_OK;
}

int launch_genotype(svt_batch* b)
{
    if (b->n_tiles == 0) return SVT_OK;
    const dim3 grid((b->n_tiles + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlock);
    const bool sso = (b->flags & SVT_FLAG_SSO_ASSOCIATION) != 0;
    if (sso) {
        if (b->lds_tables) hipLaunchKernelGGL((svt_genotype_kernel<true, true>), grid, block, b->lds_bytes, b->stream, b->args);
        else hipLaunchKernelGGL((svt_genotype_kernel<true, false>), grid, block, b->lds_bytes, b->stream, b->args);
    } else {
        if (b->lds_tables) hipLaunchKernelGGL((svt_genotype_kernel<false, true>), grid, block, b->lds_bytes, b->stream, b->args);
        else hipLaunchKernelGGL((svt_genotype_kernel<false, false>), grid, block, b->lds_bytes, b->stream, b->args);
    }
    HIP_TRY(hipGetLastError());
    return SVT_OK;
}

template <typename K>
int allow_big_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
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
    if (n >= 0xFFFFFFF0ull) return fail(SVT_ERR_INVALID, "too many units in one batch (< 2^32)");
    if (in->n_libs == 0 || in->n_libs > 256) return fail(SVT_ERR_INVALID, "n_libs must be 1..256");
    if (n && (!in->rec_offset || !in->units)) return fail(SVT_ERR_INVALID, "null unit arrays");
    if (n && in->rec_offset[0] != 0) return fail(SVT_ERR_INVALID, "rec_offset[0] must be 0");
    const uint64_t n_rec = n ? in->rec_offset[n] : 0;
    if (n_rec && !in->records) return fail(SVT_ERR_INVALID, "null records");
    if (!(in->split_weight >= 0.0) || !(in->disc_weight >= 0.0) || !std::isfinite(in->split_weight) ||
        !std::isfinite(in->disc_weight))
        return fail(SVT_ERR_INVALID, "weights must be finite and >= 0");

    int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    // ---- per-unit record counts + validation of the CSR
    std::vector<uint32_t> nrec(n);
    uint64_t max_f = 0;
    for (uint64_t u = 0; u < n; ++u) {
        if (in->rec_offset[u + 1] < in->rec_offset[u]) return fail(SVT_ERR_INVALID, "rec_offset not monotone");
        uint64_t f = in->rec_offset[u + 1] - in->rec_offset[u];
        if (f > 0x7FFFFFFFull) return fail(SVT_ERR_INVALID, "unit with too many records");
        if (in->units[u].svtype > SVT_SVTYPE_BND) return fail(SVT_ERR_INVALID, "bad svtype");
        if (in->units[u].reserved != 0 || (in->units[u].flags & ~SVT_UNIT_SKIP))
            return fail(SVT_ERR_INVALID, "unit reserved/flags bits must be 0");
        nrec[u] = (uint32_t)f;
        max_f = std::max(max_f, f);
    }

    // ---- libraries: threshold tables (host, reference expression) ------------------------
    std::vector<LibDesc> libs(in->n_libs);
    std::vector<uint32_t> hist;
    std::vector<int32_t> thr;
    for (uint32_t l = 0; l < in->n_libs; ++l) {
        const svt_library& L = in->libs[l];
        if (!L.hist || L.n_bins == 0) return fail(SVT_ERR_INVALID, "library without histogram");
        if (L.n_bins > (1u << 24)) return fail(SVT_ERR_INVALID, "histogram too wide");
        if (!std::isfinite(L.mean) || !std::isfinite(L.sd)) return fail(SVT_ERR_INVALID, "library moments not finite");
        uint64_t total = 0;
        uint32_t hmax = 0;
        for (uint32_t i = 0; i < L.n_bins; ++i) {
            total += L.hist[i];
            hmax = std::max(hmax, L.hist[i]);
            if (L.hist[i] > 0x7FFFFFFFu) return fail(SVT_ERR_INVALID, "histogram count too large");
        }
        LibDesc d{};
        d.tab_off = (uint32_t)hist.size();
        d.key_min = L.key_min;
        d.n_bins = L.n_bins;
        d.v_nondel = L.mean + L.sd * 3;  // parsers.py:873-875
        d.sd2 = 2 * L.sd;                // classic.py:339
        libs[l] = d;
        for (uint32_t i = 0; i < L.n_bins; ++i) {
            const uint32_t h1 = L.hist[i];
            hist.push_back(h1);
            int32_t t = -1;
            if (h1 > 0 && total > 0 && p_concordant_expr(h1, 0, total)) {
                // largest h2 in [0, hmax] with p > 0.5 (monotone non-increasing in h2)
                uint32_t lo = 0, hi = hmax;  // invariant: expr(lo) true
                if (p_concordant_expr(h1, hi, total)) lo = hi;
                else {
                    while (hi - lo > 1) {
                        uint32_t mid = lo + (hi - lo) / 2;
                        if (p_concordant_expr(h1, mid, total)) lo = mid; else hi = mid;
                    }
                }
                t = (int32_t)lo;
            }
            thr.push_back(t);
        }
    }
    const uint32_t total_bins = (uint32_t)hist.size();

    // ---- log10 table: n = QR + QA <= 2 * (2 * split_weight + disc_weight) * max F ----------
    const double bound = 2.0 * (2.0 * in->split_weight + in->disc_weight) * (double)max_f + 4.0;
    if (bound > 64.0 * 1024 * 1024) return fail(SVT_ERR_INVALID, "weights * records too large for the log table");
    const uint32_t n_l10 = (uint32_t)bound + 1;
    std::vector<double> l10(n_l10);
    l10[0] = 0.0;  // never read (log_choose only looks up 1..n)
    for (uint32_t i = 1; i < n_l10; ++i) l10[i] = py_log10((double)i);
    std::vector<double> pm(256);
    for (int q = 0; q < 256; ++q) pm[q] = 1.0 - std::pow(10.0, -(double)q / 10.0);  // utils.py:74-75

    // ---- tiling: sort by F inside chunks, 64 units per tile ---------------------------------
    std::vector<TileDesc> tiles_store;   // storage order
    std::vector<LaneHdr> hdr;
    std::vector<uint64_t> lane_src;
    std::vector<uint32_t> lane_nrec;
    hdr.reserve(n + kWave);
    lane_src.reserve(n + kWave);
    lane_nrec.reserve(n + kWave);
    uint64_t tiled_records = 0;
    std::vector<uint32_t> order(kChunkUnits);
    for (uint64_t c0 = 0; c0 < n; c0 += kChunkUnits) {
        const uint32_t cn = (uint32_t)std::min<uint64_t>(kChunkUnits, n - c0);
        for (uint32_t i = 0; i < cn; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.begin() + cn,
                         [&](uint32_t x, uint32_t y) { return nrec[c0 + x] > nrec[c0 + y]; });
        for (uint32_t t0 = 0; t0 < cn; t0 += kWave) {
            TileDesc td{};
            td.rec_base = tiled_records;
            td.lane_base = (uint32_t)hdr.size();
            uint32_t rows = 0;
            for (uint32_t l = 0; l < (uint32_t)kWave; ++l) {
                LaneHdr h{};
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
                } else {
                    h.unit = kPadUnit;
                }
                rows = std::max(rows, f);
                hdr.push_back(h);
                lane_src.push_back(src);
                lane_nrec.push_back(f);
            }
            td.rows = rows;
            tiled_records += (uint64_t)rows * kWave;
            tiles_store.push_back(td);
        }
    }
    if (tiles_store.size() > 0xFFFFFFF0ull / kWave) return fail(SVT_ERR_INVALID, "too many tiles");
    // dispatch order: longest tiles first (LPT) so the tail of the grid is made of short tiles
    std::vector<TileDesc> tiles_dispatch = tiles_store;
    std::stable_sort(tiles_dispatch.begin(), tiles_dispatch.end(),
                     [](const TileDesc& x, const TileDesc& y) { return x.rows > y.rows; });

    // ---- device objects ---------------------------------------------------------------------
    svt_batch* b = new (std::nothrow) svt_batch();
    if (!b) return fail(SVT_ERR_NOMEM, "out of host memory");
    b->device = device;
    b->flags = flags;
    b->n_units = n;
    b->n_records = n_rec;
    b->tiled_records = tiled_records;
    b->n_tiles = (uint32_t)tiles_store.size();

    int rc = SVT_OK;
    uint4* d_csr = nullptr;
    TileDesc* d_tiles_store = nullptr;
    uint64_t* d_lane_src = nullptr;
    uint32_t* d_lane_nrec = nullptr;
    uint32_t* d_err = nullptr;
    auto cleanup_tmp = [&]() {
        if (d_csr) (void)hipFree(d_csr);
        if (d_tiles_store) (void)hipFree(d_tiles_store);
        if (d_lane_src) (void)hipFree(d_lane_src);
        if (d_lane_nrec) (void)hipFree(d_lane_nrec);
        if (d_err) (void)hipFree(d_err);
    };
#define TRY_OR_CLEAN(expr)                                    \
    do {                                                      \
        rc = (expr);                                          \
        if (rc != SVT_OK) { cleanup_tmp(); free_batch(b); return rc; } \
    } while (0)
#define HIP_OR_CLEAN(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            cleanup_tmp(); free_batch(b);                                                   \
            return fail(SVT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
        }                                                                                   \
    } while (0)

    HIP_OR_CLEAN(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    HIP_OR_CLEAN(hipEventCreate(&b->ev0));
    HIP_OR_CLEAN(hipEventCreate(&b->ev1));
    TRY_OR_CLEAN(upload(&b->d_tiles, tiles_dispatch, b->stream));
    TRY_OR_CLEAN(upload(&d_tiles_store, tiles_store, b->stream));
    TRY_OR_CLEAN(upload(&b->d_hdr, hdr, b->stream));
    TRY_OR_CLEAN(upload(&d_lane_src, lane_src, b->stream));
    TRY_OR_CLEAN(upload(&d_lane_nrec, lane_nrec, b->stream));
    TRY_OR_CLEAN(upload(&b->d_pm, pm, b->stream));
    TRY_OR_CLEAN(upload(&b->d_l10, l10, b->stream));
    TRY_OR_CLEAN(upload(&b->d_libs, libs, b->stream));
    TRY_OR_CLEAN(upload(&b->d_hist, hist, b->stream));
    TRY_OR_CLEAN(upload(&b->d_thr, thr, b->stream));

    HIP_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_csr), std::max<uint64_t>(n_rec, 1) * sizeof(uint4)));
    if (n_rec)
        HIP_OR_CLEAN(hipMemcpyAsync(d_csr, in->records, n_rec * sizeof(uint4), hipMemcpyHostToDevice, b->stream));
    HIP_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&b->d_tiled), std::max<uint64_t>(tiled_records, 1) * sizeof(uint4)));
    HIP_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_err), sizeof(uint32_t)));
    HIP_OR_CLEAN(hipMemsetAsync(d_err, 0, sizeof(uint32_t), b->stream));

    const uint64_t n1 = std::max<uint64_t>(n, 1);
    HIP_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&b->d_gl), 3 * n1 * sizeof(double)));
    HIP_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&b->d_sq), n1 * sizeof(double)));
    HIP_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&b->d_tallies), SVT_N_TALLIES * n1 * sizeof(double)));
    HIP_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&b->d_counts), SVT_N_COUNTS * n1 * sizeof(int32_t)));
    HIP_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&b->d_gt), n1 * sizeof(int8_t)));

    // ---- re-tile on the device ------------------------------------------------------------------
    if (b->n_tiles) {
        RepackArgs ra{};
        ra.csr = d_csr;
        ra.lane_src = d_lane_src;
        ra.lane_nrec = d_lane_nrec;
        ra.tiles = d_tiles_store;
        ra.tiled = b->d_tiled;
        ra.n_tiles = b->n_tiles;
        ra.n_libs = in->n_libs;
        ra.err = d_err;
        const dim3 grid((b->n_tiles + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlock);
        hipLaunchKernelGGL(svt_repack_kernel, grid, block, 0, b->stream, ra);
        HIP_OR_CLEAN(hipGetLastError());
    }
    uint32_t err_bits = 0;
    HIP_OR_CLEAN(hipMemcpyAsync(&err_bits, d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    HIP_OR_CLEAN(hipStreamSynchronize(b->stream));
    cleanup_tmp();
    d_csr = nullptr; d_tiles_store = nullptr; d_lane_src = nullptr; d_lane_nrec = nullptr; d_err = nullptr;
    if (err_bits) {
        free_batch(b);
        std::string m = "invalid evidence records:";
        if (err_bits & 1u) m += " split bits without PRESENT;";
        if (err_bits & 2u) m += " straddle bits without HAS_PAIR;";
        if (err_bits & 4u) m += " lib index >= n_libs;";
        if (err_bits & 8u) m += " reserved/undefined bits set;";
        if (err_bits & 16u) m += " negative ospan_len;";
        return fail(SVT_ERR_INVALID, m);
    }

    // ---- kernel arguments ----------------------------------------------------------------------
    KernelArgs& a = b->args;
    a.tiled = b->d_tiled;
    a.tiles = b->d_tiles;
    a.hdr = b->d_hdr;
    a.pm = b->d_pm;
    a.l10 = b->d_l10;
    a.libs = b->d_libs;
    a.hist = b->d_hist;
    a.thr = b->d_thr;
    a.n_l10 = n_l10;
    a.n_libs = in->n_libs;
    a.total_bins = total_bins;
    a.n_tiles = b->n_tiles;
    a.l10_in_lds = n_l10 <= kMaxL10Lds ? 1u : 0u;
    a.n_units = n;
    a.gl = b->d_gl;
    a.sq = b->d_sq;
    a.tallies = b->d_tallies;
    a.counts = b->d_counts;
    a.gt = b->d_gt;
    {
        const double p_alt[2][3] = {{1e-3, 0.5, 0.9}, {1e-2, 0.2, 1 / 3.0}};  // statistics.py:26,28
        for (int d = 0; d < 2; ++d)
            for (int g = 0; g < 3; ++g) {
                a.c.lgp[d][g] = py_log10(p_alt[d][g]);
                a.c.lg1p[d][g] = py_log10(1 - p_alt[d][g]);
            }
        a.c.ln10 = std::log(10.0);
        a.c.x_uflow = find_pow10_underflow();
        a.c.split_weight = in->split_weight;
        a.c.disc_weight = in->disc_weight;
    }
    const size_t table_bytes = (size_t)total_bins * 8;
    b->lds_tables = table_bytes <= kMaxLdsTableBytes;
    const uint32_t n_l10_lds = a.l10_in_lds ? ((n_l10 + 1u) & ~1u) : 0u;
    b->lds_bytes = 256 * 8 + (size_t)n_l10_lds * 8 + (size_t)in->n_libs * sizeof(LibDesc) +
                   (b->lds_tables ? table_bytes : 0);
    b->lds_bytes = (b->lds_bytes + 15) & ~size_t(15);
    if (b->lds_bytes > 160 * 1024) { free_batch(b); return fail(SVT_ERR_INVALID, "LDS budget exceeded"); }
    TRY_OR_CLEAN(allow_big_lds(svt_genotype_kernel<false, true>, b->lds_bytes));
    TRY_OR_CLEAN(allow_big_lds(svt_genotype_kernel<true, true>, b->lds_bytes));
    TRY_OR_CLEAN(allow_big_lds(svt_genotype_kernel<false, false>, b->lds_bytes));
    TRY_OR_CLEAN(allow_big_lds(svt_genotype_kernel<true, false>, b->lds_bytes));
#undef TRY_OR_CLEAN
#undef HIP_OR_CLEAN
    *out = b;
    return SVT_OK;
}

int svt_batch_genotype(svt_batch* b, int sync)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    HIP_TRY(hipSetDevice(b->device));
    int rc = launch_genotype(b);
    if (rc != SVT_OK) return rc;
    b->have_results = true;
    if (sync) HIP_TRY(hipStreamSynchronize(b->stream));
    return SVT_OK;
}

int svt_batch_genotype_timed(svt_batch* b, int iters, float* ms_total)
{
    if (!b || !ms_total || iters <= 0) return fail(SVT_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipEventRecord(b->ev0, b->stream));
    for (int i = 0; i < iters; ++i) {
        int rc = launch_genotype(b);
        if (rc != SVT_OK) return rc;
    }
    HIP_TRY(hipEventRecord(b->ev1, b->stream));
    HIP_TRY(hipEventSynchronize(b->ev1));
    HIP_TRY(hipEventElapsedTime(ms_total, b->ev0, b->ev1));
    b->have_results = true;
    return SVT_OK;
}

int svt_batch_results(svt_batch* b, svt_results* out)
{
    if (!b || !out) return fail(SVT_ERR_INVALID, "null argument");
    if (!b->have_results) return fail(SVT_ERR_STATE, "svt_batch_genotype has not run");
    if (out->n_units != b->n_units) return fail(SVT_ERR_INVALID, "results n_units mismatch");
    HIP_TRY(hipSetDevice(b->device));
    const uint64_t n = b->n_units;
    if (n) {
        HIP_TRY(hipMemcpyAsync(out->gl, b->args.gl, 3 * n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipMemcpyAsync(out->sq, b->args.sq, n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipMemcpyAsync(out->tallies, b->args.tallies, SVT_N_TALLIES * n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipMemcpyAsync(out->counts, b->args.counts, SVT_N_COUNTS * n * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipMemcpyAsync(out->gt, b->args.gt, n * sizeof(int8_t), hipMemcpyDeviceToHost, b->stream));
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    return SVT_OK;
}

int svt_batch_device_results(svt_batch* b, svt_results* dev)
{
    if (!b || !dev) return fail(SVT_ERR_INVALID, "null argument");
    dev->n_units = b->n_units;
    dev->gl = b->args.gl;
    dev->sq = b->args.sq;
    dev->tallies = b->args.tallies;
    dev->counts = b->args.counts;
    dev->gt = b->args.gt;
    return SVT_OK;
}

int svt_batch_bind_device_results(svt_batch* b, const svt_results* dev)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    if (!dev) {
        b->args.gl = b->d_gl; b->args.sq = b->d_sq; b->args.tallies = b->d_tallies;
        b->args.counts = b->d_counts; b->args.gt = b->d_gt;
        b->bound_external = false;
        return SVT_OK;
    }
    if (dev->n_units != b->n_units) return fail(SVT_ERR_INVALID, "bound results n_units mismatch");
    if (b->n_units && (!dev->gl || !dev->sq || !dev->tallies || !dev->counts || !dev->gt))
        return fail(SVT_ERR_INVALID, "null device result pointer");
    b->args.gl = dev->gl; b->args.sq = dev->sq; b->args.tallies = dev->tallies;
    b->args.counts = dev->counts; b->args.gt = dev->gt;
    b->bound_external = true;
    b->have_results = false;
    return SVT_OK;
}

int svt_batch_bytes(const svt_batch* b, uint64_t* algorithmic, uint64_t* resident)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    if (algorithmic) *algorithmic = 16 * b->n_records + (16 + 96) * b->n_units;
    if (resident) *resident = 16 * b->tiled_records + (uint64_t)b->n_tiles * kWave * sizeof(LaneHdr);
    return SVT_OK;
}

void* svt_batch_stream(svt_batch* b) { return b ? (void*)b->stream : nullptr; }

void svt_batch_destroy(svt_batch* b) { free_batch(b); }

int svt_genotype(const svt_evidence_batch* in, svt_results* out, int device, unsigned flags)
{
    svt_batch* b = nullptr;
    int rc = svt_batch_create(in, device, flags, &b);
    if (rc != SVT_OK) return rc;
    rc = svt_batch_genotype(b, 1);
    if (rc == SVT_OK) rc = svt_batch_results(b, out);
    std::string keep = g_err;
    svt_batch_destroy(b);
    g_err = keep;
    return rc;
}

}  // extern "C"
