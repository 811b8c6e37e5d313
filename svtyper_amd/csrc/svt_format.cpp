// svt_format.cpp -- result records -> the text of VCF sample columns (include/svtyper_hip.h: svt_format_results).
//
// Host-only.  The step after the path: what the reference does per sample with a dict of FORMAT values and
// ':'.join (svtyper/classic.py:454-513, svtyper/singlesample.py:207-227,430-471,544-575; the '%0.2f' for
// floats comes from svtyper/parsers.py:391-399).  Pure formatting of what the kernels computed; the Python
// implementation of the same (svtyper_amd/results.py + vcf.Genotype.get_gt_string) stays the general path
// and is the checker of this one.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/svtyper_hip.h"
#include "svt_error.h"
#include "svt_fast_format.h"
#include "svt_host_cpus.h"

namespace {

using svt::fail;
using svt::guarded;
using svt::run_threads;

inline void put_int(std::string& s, int32_t v)
{
    char buf[16];
    char* p = buf + sizeof buf;
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *--p = '-';
    s.append(p, (size_t)(buf + sizeof buf - p));
}

inline void put_fmt(std::string& s, const char* fmt, double v)
{
    char buf[64];
    const int n = std::snprintf(buf, sizeof buf, fmt, v);
    s.append(buf, (size_t)std::max(0, std::min(n, (int)sizeof buf - 1)));
}

// '%.0f' / '%0.2f' / '%.2g' through svt_fast_format.h (exact integer arithmetic, the same digits), snprintf outside its range
inline void put_fixed(std::string& s, double v, int decimals)
{
    char buf[64];
    const int n = svt::format_fixed(buf, v, decimals);
    if (n > 0) s.append(buf, (size_t)n);
    else put_fmt(s, decimals == 0 ? "%.0f" : "%0.2f", v);
}

inline void put_g2(std::string& s, double v)
{
    char buf[64];
    const int n = svt::format_g2(buf, v);
    if (n > 0) s.append(buf, (size_t)n);
    else put_fmt(s, "%.2g", v);
}

// SQ of a called unit from its three log10 likelihoods with the HOST libm -- the very calls CPython makes for
// svtyper/classic.py:473-481: gt_sum = sum(10 ** gl), math.log(gt_sum, 10) == log(gt_sum) / log(10).  GL is
// bit-identical to the reference's, so this SQ is too (the device's own SQ goes through the GPU's exp10 / log and
// can differ in the last places: 5e-13 measured).
inline double host_sample_qual(const svt_result& r)
{
    double gt_sum = 0.0;
    for (int g = 0; g < 3; ++g) gt_sum += std::pow(10.0, r.gl[g]);
    if (!(gt_sum > 0.0)) return r.sq;           // (the device decided GT './.' against the same libm's underflow point)
    const double gt_sum_log = std::log(gt_sum) / std::log(10.0);
    return std::fabs(-10.0 * (r.gl[0] - gt_sum_log));
}

// one FORMAT value of one unit, exactly as the Python layer prints it
void put_field(std::string& s, const svt_result& r, uint8_t field, bool skipped_as_dots)
{
    const int gt = r.gt;
    if (gt == SVT_GT_SKIPPED && skipped_as_dots) {          // classic.py:282-284: only GT is set
        s += field == SVT_FMT_GT ? "./." : ".";
        return;
    }
    const bool blank = gt == SVT_GT_BLANK || gt == SVT_GT_SKIPPED;   // blank_result(): classic.py:496-513
    static const int kCount[SVT_N_FORMAT_FIELDS] = {-1, SVT_CNT_GQ, -1, -1, SVT_CNT_DP, SVT_CNT_RO, SVT_CNT_AO, SVT_CNT_QR,
                                                    SVT_CNT_QA, SVT_CNT_RS, SVT_CNT_AS, SVT_CNT_ASC, SVT_CNT_RP, SVT_CNT_AP, -1};
    switch (field) {
    case SVT_FMT_GT:
        s += gt == 0 ? "0/0" : gt == 1 ? "0/1" : gt == 2 ? "1/1" : "./.";
        return;
    case SVT_FMT_GQ:
        if (gt >= 0) put_int(s, r.counts[SVT_CNT_GQ]); else s += '.';
        return;
    case SVT_FMT_SQ:
        if (gt >= 0) put_fixed(s, r.sq, 2); else s += '.';
        return;
    case SVT_FMT_GL:
        if (blank) { s += '.'; return; }
        put_fixed(s, r.gl[0], 0); s += ',';
        put_fixed(s, r.gl[1], 0); s += ',';
        put_fixed(s, r.gl[2], 0);
        return;
    case SVT_FMT_AB: {
        const int64_t qr = blank ? 0 : r.counts[SVT_CNT_QR], qa = blank ? 0 : r.counts[SVT_CNT_QA];
        if (blank || qr + qa == 0) { s += '.'; return; }
        put_g2(s, (double)qa / (double)(qr + qa));            // classic.py:466-469
        return;
    }
    default:
        put_int(s, blank ? 0 : r.counts[kCount[field]]);
    }
}

}  // namespace

extern "C" {

static int svt_format_results_impl(const svt_result* res, uint64_t n_units, const uint8_t* fields, uint32_t n_fields,
                       int skipped_as_dots, char** text_out, uint64_t** offsets_out)
{
    if (!text_out || !offsets_out || (n_units && !res) || (n_fields && !fields)) return fail(SVT_ERR_INVALID, "null argument");
    *text_out = nullptr;
    *offsets_out = nullptr;
    for (uint32_t k = 0; k < n_fields; ++k)
        if (fields[k] >= SVT_N_FORMAT_FIELDS && fields[k] != SVT_FMT_ABSENT) return fail(SVT_ERR_INVALID, "unknown FORMAT field code");
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(svt::usable_cpus(), n_units / 2048 + 1));
    std::vector<std::string> part(nt);
    std::vector<std::vector<uint32_t>> len(nt);
    auto work = [&](unsigned t) {
        const uint64_t lo = n_units * t / nt, hi = n_units * (t + 1) / nt;
        std::string& s = part[t];
        s.reserve((size_t)(hi - lo) * 64);
        len[t].reserve((size_t)(hi - lo));
        for (uint64_t u = lo; u < hi; ++u) {
            const size_t at = s.size();
            for (uint32_t k = 0; k < n_fields; ++k) {
                if (k) s += ':';
                if (fields[k] == SVT_FMT_ABSENT) s += '.';
                else put_field(s, res[u], fields[k], skipped_as_dots != 0);
            }
            len[t].push_back((uint32_t)(s.size() - at));
        }
    };
    run_threads(nt, work);
    uint64_t total = 0;
    for (const auto& s : part) total += s.size();
    char* text = static_cast<char*>(std::malloc(std::max<uint64_t>(total, 1)));
    uint64_t* off = static_cast<uint64_t*>(std::malloc((n_units + 1) * sizeof(uint64_t)));
    if (!text || !off) {
        std::free(text);
        std::free(off);
        return fail(SVT_ERR_NOMEM, "out of host memory");
    }
    uint64_t at = 0, u = 0;
    for (unsigned t = 0; t < nt; ++t) {
        std::memcpy(text + at, part[t].data(), part[t].size());
        for (uint32_t l : len[t]) {
            off[u++] = at;
            at += l;
        }
    }
    off[n_units] = at;
    *text_out = text;
    *offsets_out = off;
    return SVT_OK;
}

int svt_format_results(const svt_result* res, uint64_t n_units, const uint8_t* fields, uint32_t n_fields, int skipped_as_dots, char** text_out, uint64_t** offsets_out)
{
    return guarded([&] { return svt_format_results_impl(res, n_units, fields, n_fields, skipped_as_dots, text_out, offsets_out); });
}

void svt_format_free(char* text, uint64_t* offsets)
{
    std::free(text);
    std::free(offsets);
}

static int svt_results_host_sq_impl(svt_result* res, uint64_t n_units)
{
    if (n_units && !res) return fail(SVT_ERR_INVALID, "null argument");
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(svt::usable_cpus(), n_units / 4096 + 1));
    run_threads(nt, [&](unsigned t) {
        const uint64_t lo = n_units * t / nt, hi = n_units * (t + 1) / nt;
        for (uint64_t u = lo; u < hi; ++u)
            if (res[u].gt >= 0) res[u].sq = host_sample_qual(res[u]);
    });
    return SVT_OK;
}

int svt_results_host_sq(svt_result* res, uint64_t n_units)
{
    return guarded([&] { return svt_results_host_sq_impl(res, n_units); });
}

}  // extern "C"
