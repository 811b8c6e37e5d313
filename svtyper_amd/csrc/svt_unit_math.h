// svt_unit_math.h -- what every kernel shares: LDS access by byte address, the per-record evidence arithmetic
// (canonical records and packed entries) and the per-unit epilogue (QR/QA, log_choose, likelihoods, decision)
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_UNIT_MATH_H
#define SVT_UNIT_MATH_H

#include "svt_device_types.h"

namespace svt {

// ------------------------------------------------------------------------------------------
// LDS access by byte address.  The tables live at fixed byte offsets of the workgroup's LDS
// (svt_device_types.h: kLds*), so an entry field that is already a byte offset becomes the operand
// of a ds_read with the table base as the instruction's immediate offset.
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) const double lds_cf64;
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef __attribute__((address_space(3))) const int32_t lds_ci32;

__device__ __forceinline__ double lds_f64(const uint32_t addr) { return *reinterpret_cast<lds_cf64*>((size_t)addr); }
__device__ __forceinline__ uint32_t lds_u32(const uint32_t addr) { return *reinterpret_cast<lds_cu32*>((size_t)addr); }
__device__ __forceinline__ int32_t lds_i32(const uint32_t addr) { return *reinterpret_cast<lds_ci32*>((size_t)addr); }
typedef __attribute__((address_space(3))) const int16_t lds_ci16;
typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
__device__ __forceinline__ int32_t lds_i16(const uint32_t addr) { return *reinterpret_cast<lds_ci16*>((size_t)addr); }
__device__ __forceinline__ uint32_t lds_u16(const uint32_t addr) { return *reinterpret_cast<lds_cu16*>((size_t)addr); }

// (byte N of e) * 8 in one VALU instruction (SDWA source select + shift): the LDS byte offset of
// prob_mapq[byte] in the double[256] table
#define SVT_BYTE_X8(N)                                                                                       \
    __device__ __forceinline__ uint32_t byte##N##_x8(const uint32_t e)                                      \
    {                                                                                                        \
        uint32_t r;                                                                                          \
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #N \
            : "=v"(r) : "v"(3u), "v"(e));                                                                    \
        return r;                                                                                            \
    }
SVT_BYTE_X8(0)
SVT_BYTE_X8(1)
SVT_BYTE_X8(2)
SVT_BYTE_X8(3)
#undef SVT_BYTE_X8

// ------------------------------------------------------------------------------------------
// evidence arithmetic
// ------------------------------------------------------------------------------------------
struct Tables {               // general mode: tables through ordinary pointers
    const double* pm;          // LDS
    const PairWeights* wtab;   // LDS
    const LibDesc* libs;       // LDS
    const Bin* bins;           // global (through L2)
};

struct Acc {
    double ref_seq, alt_seq, alt_clip, ref_span, alt_span;
    double l_ref_seq, l_alt_seq, l_alt_clip;  // sso fragment-local sums
};

// per-lane constants of the unit, hoisted out of the record loop (general mode and packed entries)
struct LaneCtx {
    uint32_t del16;       // is_DEL ? 16 : 0 (decision-table index bit)
    int32_t var_length;
    double pos_delta_d;
    bool is_del;
    // packed entries (one library, tables at fixed LDS addresses)
    uint32_t wt0, wt1;    // LDS address of w_alt[del16] / w_alt[del16 + 8] (p_concordant = 0 / 1) in the column-wise table
    uint32_t common_mq;   // mapq_a | mapq_b << 8 of the one-half-word pair entries
    uint32_t nb4, off2_4; // n_bins * 4 (byte offset of the sentinel bin); DEL ? min(var_length, n_bins) * 4 : 0x80000000
    uint32_t hist_at;     // LDS address of hist[0]
    // packed entries of several libraries (library switches, svt_entry_formats.h): the tables are read through L2
    uint32_t tab8;        // byte offset of the current library's first Bin in bins[]
    uint32_t libs_at;     // LDS address of the batch's LibDesc[n_libs]
    uint32_t lib_last;    // n_libs - 1
};

// ---- one canonical 16-byte record, any geometry (svt_stream_kernel.h: kGeneral) --------------------
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
__device__ __forceinline__ void pair_evidence(const uint32_t o, const uint32_t mq, uint32_t f3,
                                              const uint32_t lib_idx, const Tables& t, const LaneCtx& c, Acc& a)
{
    const double pm_a = t.pm[mq & 0xffu];
    const double pm_b = t.pm[(mq >> 8) & 0xffu];

    // p_concordant (parsers.py:861-882) as an integer test: with d1 = hist[o]/N fixed, the
    // reference's binary64 expression d1*0.95/(0.95*d1 + 0.05*d2) > 0.5 is monotone in
    // h2 = hist[o - v]; bins[o].thr is the largest h2 for which it still holds (found on the host
    // with the reference's own expression), -1 where hist[o] == 0 (p == 0 or ZeroDivisionError).
    int32_t thr1;
    uint32_t h2;
    {
        const LibDesc lib = t.libs[lib_idx];
        const bool small_del = c.is_del && (c.pos_delta_d < lib.sd2);
        f3 = small_del ? 0u : f3;
        const int64_t i1 = (int64_t)(int32_t)o - (int64_t)lib.key_min;
        const bool in1 = (uint64_t)i1 < (uint64_t)lib.n_bins;
        thr1 = t.bins[lib.tab_off + (in1 ? (uint32_t)i1 : lib.n_bins)].thr;
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
        h2 = t.bins[lib.tab_off + (in2 ? (uint32_t)i2 : lib.n_bins)].hist;
    }
    const bool p_conc = (int32_t)h2 <= thr1;
    const PairWeights pw = t.wtab[f3 | (p_conc ? 8u : 0u) | c.del16];
    const double pp = pm_a * pm_b;
    a.alt_span += pp * pw.w_alt;
    a.ref_span += pp * pw.w_ref;
}

// ---- packed evidence (svt_entry_formats.h describes the entries) ---------------------------------
// Pair slot (one library): a 16-byte slot is four dwords, each either two one-half-word entries that carry
// the batch's common MAPQ pair, or one wide entry (low half f3 | code << 3 | 0x8000, high half its two MAPQs).
// code4 = byte offset of thr[code] / hist[code], f3x8 = f3 << 3 (byte offset inside a decision-table column), pp = pmA * pmB.
// MULTI (several libraries): thr / hist of the entry's library come from bins[] in device memory (every library's
// n_bins + 1 Bin{thr, hist}, a few hundred KB at most: L2) -- the libraries of a batch do not fit LDS together, and a launch
// over unit ranges (the route encodes ahead of the wire) cannot group its workgroups by library window the way the
// canonical route does.  The pass over packed evidence is a few per cent of its route either way.
template <bool MULTI>
__device__ __forceinline__ void pair_tables(const uint32_t code4, const LaneCtx& c, const Bin* bins, int32_t& thr1, uint32_t& h2)
{
    if (MULTI) {
        const char* base = reinterpret_cast<const char*>(bins) + c.tab8;
        thr1 = *reinterpret_cast<const int32_t*>(base + 2u * min(code4, c.nb4));
        h2 = *reinterpret_cast<const uint32_t*>(base + 2u * min(code4 - c.off2_4, c.nb4) + 4u);
    } else {
        thr1 = lds_i32(kLdsBins + min(code4, c.nb4));
        h2 = lds_u32(c.hist_at + min(code4 - c.off2_4, c.nb4));
    }
}

// a half-word of the pair stream that is not the MAPQ half of a wide entry: a library switch (l + 1) << 3 moves the lane's
// table context to library l (clamped to the batch's libraries: slots a caller wrote itself are not trusted with addresses)
__device__ __forceinline__ void switch_library(const uint32_t h, LaneCtx& c)
{
    if ((h & 0x8007u) == 0u && h != 0u) {
        const uint32_t at = c.libs_at + min((h >> 3) - 1u, c.lib_last) * (uint32_t)sizeof(LibDesc);
        const uint32_t n_bins = lds_u32(at + 8u);
        c.tab8 = lds_u32(at) * (uint32_t)sizeof(Bin);
        c.nb4 = n_bins * 4u;
        c.off2_4 = c.is_del ? min((uint32_t)c.var_length, n_bins) * 4u : 0x80000000u;
    }
}

template <bool MULTI>
__device__ __forceinline__ void pair_eval_single(const uint32_t code4, const uint32_t f3x8, const double pp,
                                                 const LaneCtx& c, Acc& a, const Bin* bins)
{
    int32_t thr1;
    uint32_t h2;
    pair_tables<MULTI>(code4, c, bins, thr1, h2);
    const bool p_conc = (int32_t)h2 <= thr1;
    const uint32_t wa = (p_conc ? c.wt1 : c.wt0) | f3x8;
    const double w_alt = lds_f64(wa), w_ref = lds_f64(wa + kWcolRef);
    a.alt_span += pp * w_alt;
    a.ref_span += pp * w_ref;
}

template <bool MULTI>
__device__ __forceinline__ void short_pair_dword(const uint32_t e, LaneCtx& c, Acc& a, const Bin* bins)
{
    const bool wide = (e & 0x8000u) != 0u;
    const uint32_t hi = e >> 16;
    const uint32_t mq = wide ? hi : c.common_mq;
    const double pm_a = lds_f64(kLdsPm + byte0_x8(mq)), pm_b = lds_f64(kLdsPm + byte1_x8(mq));
    // (a library switch is then also evaluated as the entry it is not: no straddle bit, both weights 0, the sums receive +0.0)
    if (MULTI) switch_library(e & 0xffffu, c);
    pair_eval_single<MULTI>((e >> 1) & 0x3ffcu, (e << 3) & 0x38u, pm_a * pm_b, c, a, bins);
    if (MULTI && !wide) switch_library(hi, c);
    // the high half: a second entry with the common MAPQs -- its products pmA * pmB * {w_alt, w_ref} come ready from the
    // second decision table -- or the MAPQ bytes of the wide entry just added: then the straddle bits read as 0,
    // both table values are 0 and the sums receive +0.0
    {
        const uint32_t code4 = (e >> 17) & 0x3ffcu;
        int32_t thr1;
        uint32_t h2;
        pair_tables<MULTI>(code4, c, bins, thr1, h2);
        const bool p_conc = (int32_t)h2 <= thr1;
        const uint32_t wa = ((p_conc ? c.wt1 : c.wt0) + (kLdsWcolC - kLdsWcol)) | (wide ? 0u : (e >> 13) & 0x38u);
        a.alt_span += lds_f64(wa);
        a.ref_span += lds_f64(wa + kWcolRef);
    }
}

// Reference-read entries (classic.py:306-315): seven MAPQ pairs per row slot, byte 14 of the slot holds their
// first-of-fragment bits (bits 0..6 of f below; bit 7 is unused).  x, y = the two prob_mapq look-ups of one pair.
template <bool SSO>
__device__ __forceinline__ void ref_read_pair(const double x, const double y, const bool first, Acc& a)
{
    if (SSO) {   // singlesample.py:246-276,367-372: fragment-local sum, flushed when the next fragment starts
        a.ref_seq += first ? a.l_ref_seq : 0.0;
        a.l_ref_seq = ((first ? 0.0 : a.l_ref_seq) + x) + y;
    } else {
        a.ref_seq = (a.ref_seq + x) + y;
    }
}

template <bool SSO>
__device__ __forceinline__ void ref_read_row(const uint4 w, Acc& a)
{
    const uint32_t f = w.w >> 16;     // bit k: entry k is the first kept one of its fragment
    ref_read_pair<SSO>(lds_f64(kLdsPm + byte0_x8(w.x)), lds_f64(kLdsPm + byte1_x8(w.x)), (f & 1u) != 0u, a);
    ref_read_pair<SSO>(lds_f64(kLdsPm + byte2_x8(w.x)), lds_f64(kLdsPm + byte3_x8(w.x)), (f & 2u) != 0u, a);
    ref_read_pair<SSO>(lds_f64(kLdsPm + byte0_x8(w.y)), lds_f64(kLdsPm + byte1_x8(w.y)), (f & 4u) != 0u, a);
    ref_read_pair<SSO>(lds_f64(kLdsPm + byte2_x8(w.y)), lds_f64(kLdsPm + byte3_x8(w.y)), (f & 8u) != 0u, a);
    ref_read_pair<SSO>(lds_f64(kLdsPm + byte0_x8(w.z)), lds_f64(kLdsPm + byte1_x8(w.z)), (f & 16u) != 0u, a);
    ref_read_pair<SSO>(lds_f64(kLdsPm + byte2_x8(w.z)), lds_f64(kLdsPm + byte3_x8(w.z)), (f & 32u) != 0u, a);
    ref_read_pair<SSO>(lds_f64(kLdsPm + byte0_x8(w.w)), lds_f64(kLdsPm + byte1_x8(w.w)), (f & 64u) != 0u, a);
}

// split (alt_seq) or clip (alt_clip) candidate: the other tally receives +0.0
template <bool SSO>
__device__ __forceinline__ void candidate_pair(const double x, const double y, const bool first, const bool clip, Acc& a)
{
    const double p = (x + y) * 0.5;              // (pm(left) * L + pm(right) * R) / 2.0   (classic.py:324)
    const double ps = clip ? 0.0 : p, pc = clip ? p : 0.0;
    if (SSO) {
        const bool fs = first && !clip, fc = first && clip;
        a.alt_seq += fs ? a.l_alt_seq : 0.0;
        a.alt_clip += fc ? a.l_alt_clip : 0.0;
        a.l_alt_seq = (fs ? 0.0 : a.l_alt_seq) + ps;
        a.l_alt_clip = (fc ? 0.0 : a.l_alt_clip) + pc;
    } else {
        a.alt_seq += ps;
        a.alt_clip += pc;
    }
}

template <bool SSO>
__device__ __forceinline__ void candidate_row(const uint4 w, Acc& a)
{
    const uint32_t f = w.w >> 16, c = w.w >> 24;   // bit k: first of its fragment / clip candidate
    candidate_pair<SSO>(lds_f64(kLdsPm + byte0_x8(w.x)), lds_f64(kLdsPm + byte1_x8(w.x)), (f & 1u) != 0u, (c & 1u) != 0u, a);
    candidate_pair<SSO>(lds_f64(kLdsPm + byte2_x8(w.x)), lds_f64(kLdsPm + byte3_x8(w.x)), (f & 2u) != 0u, (c & 2u) != 0u, a);
    candidate_pair<SSO>(lds_f64(kLdsPm + byte0_x8(w.y)), lds_f64(kLdsPm + byte1_x8(w.y)), (f & 4u) != 0u, (c & 4u) != 0u, a);
    candidate_pair<SSO>(lds_f64(kLdsPm + byte2_x8(w.y)), lds_f64(kLdsPm + byte3_x8(w.y)), (f & 8u) != 0u, (c & 8u) != 0u, a);
    candidate_pair<SSO>(lds_f64(kLdsPm + byte0_x8(w.z)), lds_f64(kLdsPm + byte1_x8(w.z)), (f & 16u) != 0u, (c & 16u) != 0u, a);
    candidate_pair<SSO>(lds_f64(kLdsPm + byte2_x8(w.z)), lds_f64(kLdsPm + byte3_x8(w.z)), (f & 32u) != 0u, (c & 32u) != 0u, a);
    candidate_pair<SSO>(lds_f64(kLdsPm + byte0_x8(w.w)), lds_f64(kLdsPm + byte1_x8(w.w)), (f & 64u) != 0u, (c & 64u) != 0u, a);
}

__device__ __forceinline__ double log_choose_dev(const double* __restrict__ l10, int32_t n, int32_t k)
{
    // statistics.py:9-20 -- same loop and the same order of additions, log(i)/log(10) from the host-built
    // table; the table reads of four iterations are issued together so the chain of dependent adds
    // does not wait for one LDS round trip per term
    double r = 0.0;
    if (k * 2 > n) k = n - k;
    int32_t d = 1;
    for (; d + 3 <= k; d += 4) {
        const double a0 = l10[n], a1 = l10[n - 1], a2 = l10[n - 2], a3 = l10[n - 3];
        const double b0 = l10[d], b1 = l10[d + 1], b2 = l10[d + 2], b3 = l10[d + 3];
        r += a0; r -= b0;
        r += a1; r -= b1;
        r += a2; r -= b2;
        r += a3; r -= b3;
        n -= 4;
    }
    for (; d <= k; ++d) {
        r += l10[n];
        r -= l10[d];
        n -= 1;
    }
    return r;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 pack2d(double x, double y)
{
    const uint64_t a = (uint64_t)__double_as_longlong(x), b = (uint64_t)__double_as_longlong(y);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}

// ------------------------------------------------------------------------------------------
// per-unit epilogue: zeroing rules -> QR/QA -> bayes_gt -> GT/GQ/SQ -> the eight 16-byte pieces of the
// 128-byte result record (classic.py:425-513).  Shared by every genotype kernel, so all device layouts
// produce the same bits.  The host-built log(i)/log(10) table is read from LDS (l10_lds: its first l10_lds_entries entries) or through L2 (l10_global).
// RULES = false / BLANKS = false: the seam form bayesian_genotype(counts) (singlesample.py:406-473), which takes the
// counts as they are -- its callers apply the zeroing rules and pick the blank result before calling it.
// ------------------------------------------------------------------------------------------
template <bool RULES = true, bool BLANKS = true>
__device__ __forceinline__ void unit_epilogue(const Acc& acc, const uint32_t svtype, const uint32_t uflags, const GtConsts& c,
                                              const double* l10_lds, const double* __restrict__ l10_global, const uint32_t l10_lds_entries,
                                              uint4 (&piece)[8])
{
    double ref_seq = acc.ref_seq, alt_seq = acc.alt_seq, alt_clip = acc.alt_clip,
           ref_span = acc.ref_span, alt_span = acc.alt_span;

    // ---- zeroing rules (classic.py:425-435)
    if (RULES) {
        if ((alt_seq + alt_clip) < 0.5 && alt_span >= 1.0) { alt_seq = 0.0; alt_clip = 0.0; ref_seq = 0.0; }
        if (alt_span < 0.5 && (alt_seq + alt_clip) >= 1.0) { alt_span = 0.0; ref_span = 0.0; }
        if (alt_span + alt_seq == 0.0 && alt_clip > 0.0) alt_clip = 0.0;
    }

    int32_t cnt[SVT_N_COUNTS];
#pragma unroll
    for (int i = 0; i < SVT_N_COUNTS; ++i) cnt[i] = 0;
    double gl[3] = {0.0, 0.0, 0.0};
    double sq = 0.0;
    int32_t gt;

    const bool skipped = BLANKS && (uflags & SVT_UNIT_SKIP) != 0;
    const bool evidence = !BLANKS || (ref_seq + alt_seq + ref_span + alt_span + alt_clip) > 0.0;  // classic.py:437
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
        const int32_t QR = (int32_t)(c.split_weight * ref_seq) + (int32_t)(c.disc_weight * ref_span);      // :443
        const int32_t QA = (int32_t)(c.split_weight * alt_splitters) + (int32_t)(c.disc_weight * alt_span); // :444
        // bayes_gt (statistics.py:23-37)
        const int32_t total = QR + QA;
        double log_combo;
        // two call sites, ds_read vs global_load: log_choose reads l10[1..k] and l10[n-k+1..n], so `total` is the largest index it
        // touches -- a unit whose counts stay below the part of the table that sits in LDS never leaves it
        if ((uint32_t)total < l10_lds_entries) log_combo = log_choose_dev(l10_lds, total, QA);
        else log_combo = log_choose_dev(l10_global, total, QA);
#pragma unroll
        for (int g = 0; g < 3; ++g)
            gl[g] = (log_combo + (double)QA * c.lgp[is_dup][g]) + (double)QR * c.lg1p[is_dup][g];

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
        if (gl_best >= c.x_uflow) {
            // 10**gl: exp10 (no logarithm inside, a third of pow's instructions) wherever its last-place error
            // cannot show -- with the largest term far above the subnormal range every term is either accurate
            // to an ulp or negligible beside it.  Sums near the underflow point keep pow, whose rounding of
            // subnormal results is the one the parity tests pinned against the host libm.
            double gt_sum = 0.0;
            if (gl_best >= -290.0) {
#pragma unroll
                for (int g = 0; g < 3; ++g) gt_sum += exp10(gl[g]);
            } else {
#pragma unroll
                for (int g = 0; g < 3; ++g) gt_sum += pow(10.0, gl[g]);
            }
            const double gt_sum_log = log(gt_sum) / c.ln10;                       // :480
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

    // ---- one 128-byte result record per unit, scattered back to the unit's original position
    const uint64_t t4 = (uint64_t)__double_as_longlong(alt_span);
    const uint4 out_piece[8] = {
        pack2d(gl[0], gl[1]),
        pack2d(gl[2], sq),
        pack2d(ref_seq, alt_seq),
        pack2d(alt_clip, ref_span),
        make_uint4((uint32_t)t4, (uint32_t)(t4 >> 32), (uint32_t)cnt[0], (uint32_t)cnt[1]),
        make_uint4((uint32_t)cnt[2], (uint32_t)cnt[3], (uint32_t)cnt[4], (uint32_t)cnt[5]),
        make_uint4((uint32_t)cnt[6], (uint32_t)cnt[7], (uint32_t)cnt[8], (uint32_t)cnt[9]),
        make_uint4((uint32_t)cnt[10], (uint32_t)gt & 0xffu, 0u, 0u),
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) piece[i] = out_piece[i];
}

}  // namespace svt

#endif  // SVT_UNIT_MATH_H
