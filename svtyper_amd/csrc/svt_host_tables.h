// svt_host_tables.h -- host-side look-up tables (same libm calls CPython makes)
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_HOST_TABLES_H
#define SVT_HOST_TABLES_H

#include "svt_device_types.h"

namespace svt {

// ------------------------------------------------------------------------------------------
// host-side table construction (same libm calls CPython makes)
// ------------------------------------------------------------------------------------------

// parsers.py:861-882 for counts (h1, h2) of a library with N samples
inline bool p_concordant_expr(uint32_t h1, uint32_t h2, uint64_t n_total)
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

inline double py_log10(double x) { return std::log(x) / std::log(10.0); }  // math.log(x, 10)

// smallest double x with pow(10.0, x) > 0 under this libm (CPython: 10 ** x)
inline double find_pow10_underflow()
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

inline void fill_gt_consts(GtConsts& c, double split_weight, double disc_weight)
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
    std::vector<Bin> bins;           // per library: n_bins {threshold, count} + sentinel {-1, 0}, both as ranks (build_tables)
    std::vector<PairWeights> wtab;   // 32
    std::vector<double> pm;          // 256
    std::vector<double> l10;
    bool fast_geometry = true;       // 32-bit index math + "non-DEL key never integral" valid?
    bool narrow_bins = true;         // every library's ranks fit 16 bits (the streaming kernel's LDS tables)
};

inline int build_tables(const svt_evidence_batch* in, uint64_t max_records_per_unit, HostTables& T)
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
        d.tab_off = (uint32_t)T.bins.size();
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
            T.bins.push_back(Bin{t, h1});
        }
        // out-of-range sentinel: Counter miss -> count 0; hist[o] == 0 -> never concordant
        T.bins.push_back(Bin{-1, 0u});
        // The kernels only ever ask `hist[a] <= thr[b]` inside one library: replace counts and thresholds by their
        // ranks among the library's values (an order-preserving map, thr = -1 "never" stays -1).  2 n_bins + 1 values
        // at most, so a library of up to 16 383 bins fits 16-bit tables in LDS (svt_stream_kernel.h).
        {
            Bin* lb = T.bins.data() + d.tab_off;
            std::vector<uint32_t> vals;
            vals.reserve(2 * (size_t)L.n_bins + 2);
            for (uint32_t i = 0; i <= L.n_bins; ++i) {
                vals.push_back(lb[i].hist);
                if (lb[i].thr >= 0) vals.push_back((uint32_t)lb[i].thr);
            }
            std::sort(vals.begin(), vals.end());
            vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
            auto rank = [&](const uint32_t x) { return (uint32_t)(std::lower_bound(vals.begin(), vals.end(), x) - vals.begin()); };
            for (uint32_t i = 0; i <= L.n_bins; ++i) {
                lb[i].hist = rank(lb[i].hist);
                if (lb[i].thr >= 0) lb[i].thr = (int32_t)rank((uint32_t)lb[i].thr);
            }
            if (vals.size() > 32767) T.narrow_bins = false;
        }
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


}  // namespace svt

#endif  // SVT_HOST_TABLES_H
