// svt_genotype_kernel.h -- evidence arithmetic and the genotype kernel (the hot path)
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_GENOTYPE_KERNEL_H
#define SVT_GENOTYPE_KERNEL_H

#include "svt_device_types.h"

namespace svt {

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
    uint32_t off2;        // compact layout, kSingleLds: DEL ? min(var_length, n_bins) : 0x80000000
    uint32_t lib_min;     // compact layout, kMultiLds: first library of the lane's unit
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

// ---- compact layout (svt_prepare_kernels.h describes the entries) -------------------------------
// Pair entry: the two table indices come from `code` by clamping; the look-ups, the p_concordant
// decision and the sums are the ones of pair_evidence above.
template <int MODE>
__device__ __forceinline__ void pair_entry(const uint32_t e, const Tables& t, const LaneCtx& c, Acc& a)
{
    const uint32_t code = e & ((1u << kCodeBits) - 1u);
    const uint32_t f3 = (e >> kCodeBits) & 7u;
    double pm_a, pm_b;
    int32_t thr1;
    uint32_t h2;
    if (MODE == kSingleLds) {
        pm_a = t.pm[(e >> 16) & 0xffu];
        pm_b = t.pm[e >> 24];
        thr1 = t.thr[min(code, c.nb)];
        h2 = t.hist[min(code - c.off2, c.nb)];
    } else {
        pm_a = t.pm[(e >> 16) & 0x7fu];
        pm_b = t.pm[(e >> 23) & 0x7fu];
        const LibDesc lib = t.libs[c.lib_min + (e >> 30) - c.lib_lo];
        const uint32_t off2 = c.is_del ? min((uint32_t)c.var_length, lib.n_bins) : 0x80000000u;
        const uint32_t base = lib.tab_off - c.bin_lo;
        thr1 = t.thr[base + min(code, lib.n_bins)];
        h2 = t.hist[base + min(code - off2, lib.n_bins)];
    }
    const bool p_conc = (int32_t)h2 <= thr1;
    const PairWeights pw = t.wtab[f3 | (p_conc ? 8u : 0u) | c.del16];
    const double pp = pm_a * pm_b;
    a.alt_span += pp * pw.w_alt;
    a.ref_span += pp * pw.w_ref;
}

// Weight entry of one kind (classic.py:306-328): the other two sums receive +0.0.
template <bool SSO>
__device__ __forceinline__ void weight_entry(const uint32_t e, const Tables& t, Acc& a)
{
    const double x = t.pm[e & 0xffu];
    const double y = t.pm[(e >> 8) & 0xffu];
    const uint32_t kind = (e >> 16) & 3u;
    const double xr = kind == 0u ? x : 0.0, yr = kind == 0u ? y : 0.0;
    const double p = (x + y) * 0.5;                       // (pm(left) * L + pm(right) * R) / 2.0
    const double ps = kind == 1u ? p : 0.0, pc = kind == 2u ? p : 0.0;
    if (SSO) {
        // singlesample.py:246-276,367-372: fragment-local sums, added to the site totals when the
        // next fragment (with evidence of this kind) starts
        const bool first = (e & (1u << 18)) != 0u;
        const bool f0 = first && kind == 0u, f1 = first && kind == 1u, f2 = first && kind == 2u;
        a.ref_seq += f0 ? a.l_ref_seq : 0.0;
        a.alt_seq += f1 ? a.l_alt_seq : 0.0;
        a.alt_clip += f2 ? a.l_alt_clip : 0.0;
        a.l_ref_seq = ((f0 ? 0.0 : a.l_ref_seq) + xr) + yr;
        a.l_alt_seq = (f1 ? 0.0 : a.l_alt_seq) + ps;
        a.l_alt_clip = (f2 ? 0.0 : a.l_alt_clip) + pc;
    } else {
        a.ref_seq = (a.ref_seq + xr) + yr;
        a.alt_seq += ps;
        a.alt_clip += pc;
    }
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
template <bool SSO, int MODE, bool COMPACT>
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
        c.off2 = c.is_del ? min((uint32_t)h.var_length, a.lib0.n_bins) : 0x80000000u;
        c.lib_min = (h.packed >> 16) & 0xffu;
    }

    Acc acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

    // ---- stream the tile: row j is one contiguous 1 KiB line for the wave
    if (COMPACT) {
        stream_rows<SVT_GROUP_A>(a.tiled + td.base_a + lane, td.rows_a, [&](const uint4 w) {
            pair_entry<MODE>(w.x, t, c, acc);
            pair_entry<MODE>(w.y, t, c, acc);
            pair_entry<MODE>(w.z, t, c, acc);
            pair_entry<MODE>(w.w, t, c, acc);
        });
        stream_rows<SVT_GROUP_B>(a.tiled + td.base_b + lane, td.rows_b, [&](const uint4 w) {
            weight_entry<SSO>(w.x, t, acc);
            weight_entry<SSO>(w.y, t, acc);
            weight_entry<SSO>(w.z, t, acc);
            weight_entry<SSO>(w.w, t, acc);
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


}  // namespace svt

#endif  // SVT_GENOTYPE_KERNEL_H
