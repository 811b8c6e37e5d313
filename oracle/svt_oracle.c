/*
 * svt_oracle.c -- CPU restatement (plain C, sequential IEEE-754 binary64) of the
 * SVTyper v0.7.1 likelihood hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may load it.  The product path
 * (svtyper_amd -> libsvtyper_hip.so) never calls into this file.
 *
 * Parity status: PINNED.  Checked in tests/test_oracle_golden.py against golden
 * vectors produced by importing the reference itself (tests/golden/make_golden.py,
 * run in the dev container): the bayes_gt/log_choose/genotype grid, the 211 sites
 * of the reference's own fixture (tests/data) and synthetic fake-read sites.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference checkout).  Arithmetic follows CPython's evaluation order exactly:
 * left-to-right binary64, libm pow/log as CPython calls them, no FMA contraction
 * (build with -ffp-contract=off).
 */
#include "svt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* svtyper/utils.py:74-75  prob_mapq: 1 - 10 ** (-read.mapping_quality / 10.0) */
double svt_oracle_prob_mapq(int mapq)
{
    return 1.0 - pow(10.0, -(double)mapq / 10.0);
}

/* CPython math.log(x, 10) == log(x) / log(10)  (NOT log10) -- used by
 * svtyper/statistics.py:16-17,33-35 and classic.py:480 */
static double py_log10(double x)
{
    return log(x) / log(10.0);
}

/* svtyper/statistics.py:9-20  log_choose */
double svt_oracle_log_choose(int64_t n, int64_t k)
{
    double r = 0.0;
    if (k * 2 > n) k = n - k;                 /* :12-13 */
    for (int64_t d = 1; d <= k; ++d) {        /* :15 */
        r += py_log10((double)n);             /* :16 */
        r -= py_log10((double)d);             /* :17 */
        n -= 1;                               /* :18 */
    }
    return r;
}

/* svtyper/statistics.py:23-37  bayes_gt */
void svt_oracle_bayes_gt(int64_t ref, int64_t alt, int is_dup, double out[3])
{
    double p_alt[3];
    if (is_dup) { p_alt[0] = 1e-2; p_alt[1] = 0.2; p_alt[2] = 1 / 3.0; } /* :26 */
    else        { p_alt[0] = 1e-3; p_alt[1] = 0.5; p_alt[2] = 0.9; }     /* :28 */
    int64_t total = ref + alt;                                            /* :30 */
    double log_combo = svt_oracle_log_choose(total, alt);                 /* :31 */
    for (int g = 0; g < 3; ++g) {                                         /* :33-35 */
        double a = (double)alt * py_log10(p_alt[g]);
        double b = (double)ref * py_log10(1 - p_alt[g]);
        out[g] = (log_combo + a) + b;
    }
}

/* Counter lookup of svtyper/parsers.py:878: dens[key] with dens built at :579-583
 * as float(hist[i]) / countRecords(hist); a missing key yields 0. */
static double dens_lookup(const svt_library* lib, uint64_t n_total, int64_t key)
{
    int64_t i = key - (int64_t)lib->key_min;
    if (i < 0 || i >= (int64_t)lib->n_bins) return 0.0;
    uint32_t h = lib->hist[i];
    if (h == 0) return 0.0; /* key absent from the Counter (or explicit 0 count) */
    return (double)h / (double)n_total;
}

static uint64_t lib_total(const svt_library* lib)
{
    uint64_t n = 0;
    for (uint32_t i = 0; i < lib->n_bins; ++i) n += lib->hist[i];
    return n;
}

/* svtyper/parsers.py:861-882  SamFragment.p_concordant -> bool.
 * has_var_length == 0 restates `var_length is None` (:874-875): the key becomes
 * the FLOAT ospan_length - (mean + sd*3), which only matches an integer Counter
 * key when it is integral.  ZeroDivisionError -> p = None -> (None > 0.5) is
 * False under Python 2 (:879-882). */
int svt_oracle_p_concordant(const svt_library* lib, uint64_t n_total,
                            int32_t ospan_length, int has_var_length,
                            int32_t var_length)
{
    const double disc_prior = 0.05;            /* :863 */
    const double conc_prior = 1 - disc_prior;  /* :864 */
    double d1 = dens_lookup(lib, n_total, (int64_t)ospan_length);
    double d2;
    if (has_var_length) {
        d2 = dens_lookup(lib, n_total, (int64_t)ospan_length - (int64_t)var_length);
    } else {
        double v = lib->mean + lib->sd * 3;    /* :873-875 */
        double key = (double)ospan_length - v;
        if (key == floor(key) && fabs(key) < 9.0e15)
            d2 = dens_lookup(lib, n_total, (int64_t)key);
        else
            d2 = 0.0;
    }
    double den = conc_prior * d1 + disc_prior * d2;   /* :878 */
    if (den == 0.0) return 0;                         /* ZeroDivisionError -> None -> False */
    double p = d1 * conc_prior / den;
    return p > 0.5;                                   /* :882 */
}

/* The per-fragment loop + zeroing rules:
 *   classic: svtyper/classic.py:286-435
 *   sso    : svtyper/singlesample.py:246-404 (fragment-local sums, :367-378)
 * out[SVT_TAL_*] receives the five tallies AFTER the zeroing rules. */
void svt_oracle_tally(const svt_unit* unit, const svt_record* recs, uint64_t n_recs,
                      const svt_library* libs, const uint64_t* lib_totals,
                      const double pmapq[256], int sso, double out[SVT_N_TALLIES])
{
    double ref_span = 0, alt_span = 0, ref_seq = 0, alt_seq = 0, alt_clip = 0;
    /* sso: fragment-local accumulators (singlesample.py:247), carried across
     * SVT_REC_CONTINUATION records of one fragment */
    double l_ref_seq = 0, l_alt_seq = 0, l_alt_clip = 0;
    const int is_del = unit->svtype == SVT_SVTYPE_DEL;

    for (uint64_t j = 0; j < n_recs; ++j) {
        const svt_record* r = &recs[j];
        const uint32_t f = r->flags;
        const svt_library* lib = &libs[SVT_REC_LIB(f)];

        if (sso && !(f & SVT_REC_CONTINUATION)) {
            /* singlesample.py:370-372: site totals += fragment-local sums */
            if (j > 0) { ref_seq += l_ref_seq; alt_seq += l_alt_seq; alt_clip += l_alt_clip; }
            l_ref_seq = 0; l_alt_seq = 0; l_alt_clip = 0;
        }

        /* --- reference split-read evidence: classic.py:306-311.  rs_x is the read's MAPQ when
         * is_ref_seq holds and 0 otherwise; prob_mapq(0) == 0.0 so the add is a no-op then. */
        if (sso) { l_ref_seq += pmapq[r->rs_a]; l_ref_seq += pmapq[r->rs_b]; }
        else     { ref_seq += pmapq[r->rs_a];   ref_seq += pmapq[r->rs_b]; }

        /* --- alternate split-read evidence: classic.py:317-328
         *     p_alt = (prob_mapq(left) * L + prob_mapq(right) * R) / 2.0  with the booleans
         *     L, R already folded into the bytes (MAPQ 0 where False). */
        {
            double p_seq = (pmapq[r->seq_l] + pmapq[r->seq_r]) / 2.0;      /* not is_soft_clip */
            double p_clip = (pmapq[r->clip_l] + pmapq[r->clip_r]) / 2.0;   /* is_soft_clip     */
            if (sso) { l_alt_seq += p_seq; l_alt_clip += p_clip; }
            else     { alt_seq += p_seq;   alt_clip += p_clip; }
        }

        /* --- paired-end evidence: classic.py:339-408 --- */
        const int small_del = is_del && ((double)unit->pos_delta < 2 * lib->sd); /* :339,383 */
        const int alt_straddle = !small_del && (f & SVT_REC_ALT_STRADDLE);       /* :339-357 */
        const double pm_a = pmapq[r->mapq_a], pm_b = pmapq[r->mapq_b];
        const uint32_t li = SVT_REC_LIB(f);
        if (alt_straddle) {                                                      /* :359 */
            if (is_del) {                                                        /* :360-364 */
                int p_conc = svt_oracle_p_concordant(lib, lib_totals[li], r->ospan_len, 1,
                                                     unit->var_length);
                alt_span += (1 - p_conc) * pm_a * pm_b;
            } else {
                alt_span += pm_a * pm_b;                                         /* :376-377 */
            }
        }
        const int rs_a = !small_del && (f & SVT_REC_REF_STRADDLE_A);             /* :383-396 */
        const int rs_b = !small_del && (f & SVT_REC_REF_STRADDLE_B);
        if (rs_a || rs_b) {                                                      /* :398 */
            if (!(rs_a && rs_b) || is_del) {                                     /* :401 */
                int p_conc = svt_oracle_p_concordant(lib, lib_totals[li], r->ospan_len,
                                                     is_del, unit->var_length);  /* :402 */
                double p_reference = p_conc * pm_a * pm_b;                       /* :404 */
                ref_span += (rs_a + rs_b) * p_reference / 2;                     /* :405 */
            }
        }
    }
    if (sso && n_recs > 0) { ref_seq += l_ref_seq; alt_seq += l_alt_seq; alt_clip += l_alt_clip; }

    /* --- zeroing rules: classic.py:425-435 / singlesample.py:384-393 --- */
    if ((alt_seq + alt_clip) < 0.5 && alt_span >= 1) { alt_seq = 0; alt_clip = 0; ref_seq = 0; }
    if (alt_span < 0.5 && (alt_seq + alt_clip) >= 1) { alt_span = 0; ref_span = 0; }
    if (alt_span + alt_seq == 0 && alt_clip > 0) alt_clip = 0;

    out[SVT_TAL_REF_SEQ] = ref_seq;
    out[SVT_TAL_ALT_SEQ] = alt_seq;
    out[SVT_TAL_ALT_CLIP] = alt_clip;
    out[SVT_TAL_REF_SPAN] = ref_span;
    out[SVT_TAL_ALT_SPAN] = alt_span;
}

/* classic.py:437-513 / singlesample.py:406-473 + :494-496: tallies -> result */
void svt_oracle_genotype(const double t[SVT_N_TALLIES], int svtype, double split_weight,
                         double disc_weight, double gl[3], double* sq,
                         int32_t counts[SVT_N_COUNTS], int8_t* gt)
{
    const double ref_seq = t[SVT_TAL_REF_SEQ], alt_seq = t[SVT_TAL_ALT_SEQ],
                 alt_clip = t[SVT_TAL_ALT_CLIP], ref_span = t[SVT_TAL_REF_SPAN],
                 alt_span = t[SVT_TAL_ALT_SPAN];
    memset(counts, 0, sizeof(int32_t) * SVT_N_COUNTS);
    gl[0] = gl[1] = gl[2] = 0.0;
    *sq = 0.0;
    /* classic.py:437 `if ref_seq + alt_seq + ref_span + alt_span + alt_clip > 0`;
     * singlesample.py:494-496 `total == 0` -- both mean "all five are zero". */
    if (!(ref_seq + alt_seq + ref_span + alt_span + alt_clip > 0)) {
        *gt = SVT_GT_BLANK;                      /* classic.py:496-513 */
        counts[SVT_CNT_GQ] = -1;
        return;
    }
    const int is_dup = svtype == SVT_SVTYPE_DUP;                       /* :439 */
    const double alt_splitters = alt_seq + alt_clip;                   /* :442 */
    const int64_t QR = (int64_t)(split_weight * ref_seq) + (int64_t)(disc_weight * ref_span);      /* :443 */
    const int64_t QA = (int64_t)(split_weight * alt_splitters) + (int64_t)(disc_weight * alt_span); /* :444 */
    svt_oracle_bayes_gt(QR, QA, is_dup, gl);                           /* :445 */

    /* :446 sorted(enumerate, key=value, reverse=True)[0:2] -- stable, so ties keep
     * the lower genotype index first */
    int order[3] = {0, 1, 2};
    for (int i = 1; i < 3; ++i) {             /* stable insertion sort, descending */
        int o = order[i], k = i;
        while (k > 0 && gl[order[k - 1]] < gl[o]) { order[k] = order[k - 1]; --k; }
        order[k] = o;
    }
    const int best = order[0], second = order[1];

    counts[SVT_CNT_QR] = (int32_t)QR;                                             /* :458 */
    counts[SVT_CNT_QA] = (int32_t)QA;                                             /* :459 */
    counts[SVT_CNT_DP] = (int32_t)(ref_seq + alt_seq + alt_clip + ref_span + alt_span); /* :455 */
    counts[SVT_CNT_RO] = (int32_t)(ref_seq + ref_span);                           /* :456 */
    counts[SVT_CNT_AO] = (int32_t)(alt_seq + alt_clip + alt_span);                /* :457 */
    counts[SVT_CNT_RS] = (int32_t)ref_seq;                                        /* :461 */
    counts[SVT_CNT_AS] = (int32_t)alt_seq;                                        /* :462 */
    counts[SVT_CNT_ASC] = (int32_t)alt_clip;                                      /* :463 */
    counts[SVT_CNT_RP] = (int32_t)ref_span;                                       /* :464 */
    counts[SVT_CNT_AP] = (int32_t)alt_span;                                       /* :465 */

    double gt_sum = 0;                                                            /* :473-478 */
    for (int g = 0; g < 3; ++g) gt_sum += pow(10.0, gl[g]);
    if (gt_sum > 0) {                                                             /* :479 */
        double gt_sum_log = py_log10(gt_sum);                                     /* :480 */
        *sq = fabs(-10 * (gl[0] - gt_sum_log));                                   /* :481 */
        double phred_gq = -10 * (gl[second] - gl[best]);                          /* :482 */
        if (phred_gq > 200) phred_gq = 200;      /* min(x, 200) */
        counts[SVT_CNT_GQ] = (int32_t)phred_gq;                                   /* :483 */
        *gt = (int8_t)best;                                                       /* :486-491 */
    } else {
        counts[SVT_CNT_GQ] = -1;                                                  /* :493-495 */
        *gt = SVT_GT_MISSING;
    }
}

int svt_oracle_batch(const svt_evidence_batch* in, svt_result* out, unsigned flags,
                     int n_threads)
{
    const uint64_t n = in->n_units;
    const int sso = (flags & SVT_FLAG_SSO_ASSOCIATION) != 0;
    double pmapq[256];
    for (int q = 0; q < 256; ++q) pmapq[q] = svt_oracle_prob_mapq(q);
    uint64_t* totals = (uint64_t*)malloc(sizeof(uint64_t) * (in->n_libs ? in->n_libs : 1));
    if (!totals) return -2;
    for (uint32_t l = 0; l < in->n_libs; ++l) totals[l] = lib_total(&in->libs[l]);
#ifdef _OPENMP
    const int nt = n_threads > 0 ? n_threads : omp_get_max_threads();
#else
    const int nt = 1;
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt)
    for (int64_t u = 0; u < (int64_t)n; ++u) {
        svt_result* r = &out[u];
        const svt_unit* unit = &in->units[u];
        memset(r, 0, sizeof *r);
        if (unit->flags & SVT_UNIT_SKIP) {
            r->counts[SVT_CNT_GQ] = -1;
            r->gt = SVT_GT_SKIPPED;
        } else {
            svt_oracle_tally(unit, in->records + in->rec_offset[u],
                             in->rec_offset[u + 1] - in->rec_offset[u], in->libs, totals,
                             pmapq, sso, r->tallies);
            svt_oracle_genotype(r->tallies, unit->svtype, in->split_weight, in->disc_weight,
                                r->gl, &r->sq, r->counts, &r->gt);
        }
    }
    free(totals);
    return 0;
}

int svt_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
