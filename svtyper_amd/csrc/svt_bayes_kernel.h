// svt_bayes_kernel.h -- array form of statistics.bayes_gt / log_choose
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_BAYES_KERNEL_H
#define SVT_BAYES_KERNEL_H

#include "svt_unit_math.h"

namespace svt {

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
// bayesian_genotype seam kernel (singlesample.py:406-473): the five counts of an item as its caller has them ->
// the full result record.  One item per thread; no zeroing rule, no blank shortcut (see unit_epilogue).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void svt_counts_kernel(const double* __restrict__ counts, const uint8_t* __restrict__ is_dup,
                                                            uint64_t n, const double* __restrict__ l10, const GtConsts c,
                                                            svt_result* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double* t = counts + 5 * i;   // SVT_TAL_* order: ref_seq, alt_seq, alt_clip, ref_span, alt_span
    const Acc acc = {t[0], t[1], t[2], t[3], t[4], 0.0, 0.0, 0.0};
    uint4 piece[8];
    unit_epilogue<false, false>(acc, is_dup[i] ? SVT_SVTYPE_DUP : SVT_SVTYPE_DEL, 0u, c, l10, l10, 0u, piece);
    uint4* dst = reinterpret_cast<uint4*>(out + i);
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = piece[k];
}

// ------------------------------------------------------------------------------------------
// QUAL of a site over its samples (classic.py:216-217,485,498): a running binary64 sum of SQ in -B
// order, reset to 0 by a sample without evidence, untouched by a skipped or './.' sample.  One site
// per thread, samples in order; units are site-major (unit = site * n_samples + sample).
// ------------------------------------------------------------------------------------------
// `res`: result records of `stride` bytes (svt_result, or svt_result96 under SVT_FLAG_RESULT96: SQ sits at the same offset in both,
// GT at `gt_at`)
__global__ __launch_bounds__(kBlock) void svt_site_qual_kernel(const unsigned char* __restrict__ res, uint32_t stride, uint32_t gt_at,
                                                               uint32_t n_samples, const double* __restrict__ initial,
                                                               double* __restrict__ qual, uint64_t n_sites)
{
    const uint64_t site = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (site >= n_sites) return;
    double q = initial ? initial[site] : 0.0;
    const unsigned char* r = res + site * n_samples * stride;
    for (uint32_t s = 0; s < n_samples; ++s, r += stride) {
        const int gt = *reinterpret_cast<const int8_t*>(r + gt_at);
        if (gt >= 0) q += *reinterpret_cast<const double*>(r + offsetof(svt_result, sq));   // classic.py:485
        else if (gt == SVT_GT_BLANK) q = 0.0;                                               // classic.py:498
    }
    qual[site] = q;
}

// The same over TAGGED 96-byte records (SVT_FLAG_RESULT96): they lie in the order the pass's workgroups finished them, each
// carrying the index it belongs at (svt_result96.unit; site-major after svt_batch_result_order), so a site's samples are not
// neighbours.  Two launches instead of 128 bytes per unit over PCIe and a host loop: (1) every slot puts the two fields QUAL
// needs -- SQ and GT, 16 bytes -- where its tag says (records read as whole lines, one 16-byte store per unit); (2) the running
// sum of classic.py:485,498 over a site's entries, in sample order: the same operands in the same order as the kernel above.
struct QualEntry { double sq; int64_t gt; };
static_assert(sizeof(QualEntry) == 16, "one 16-byte store per unit");

__global__ __launch_bounds__(kBlock) void svt_site_qual_scatter_kernel(const svt_result96* __restrict__ rec, uint64_t n_slots, uint64_t n_units,
                                                                       QualEntry* __restrict__ entries, uint32_t* __restrict__ err)
{
    const uint64_t s = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_slots) return;
    const uint32_t unit = rec[s].unit;
    if (unit == SVT_NO_UNIT) return;                       // padding of a workgroup's last tile
    if (unit >= n_units) { atomicOr(err, 1u); return; }    // (not a record this batch's pass wrote)
    QualEntry e;
    e.sq = rec[s].sq;
    e.gt = rec[s].gt;
    entries[unit] = e;
}

__global__ __launch_bounds__(kBlock) void svt_site_qual_entries_kernel(const QualEntry* __restrict__ entries, uint32_t n_samples,
                                                                       const double* __restrict__ initial, double* __restrict__ qual, uint64_t n_sites)
{
    const uint64_t site = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (site >= n_sites) return;
    double q = initial ? initial[site] : 0.0;
    const QualEntry* e = entries + site * n_samples;
    for (uint32_t s = 0; s < n_samples; ++s) {
        const QualEntry x = e[s];
        if (x.gt >= 0) q += x.sq;                          // classic.py:485
        else if (x.gt == SVT_GT_BLANK) q = 0.0;            // classic.py:498
    }
    qual[site] = q;
}

}  // namespace svt

#endif  // SVT_BAYES_KERNEL_H
