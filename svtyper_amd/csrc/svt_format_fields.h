// svt_format_fields.h -- one FORMAT value of one unit as VCF text: shared by svt_format.cpp (sample columns of a chunk) and
// svt_vcf.cpp (whole output lines of a chunk).  What the reference prints per sample from a dict of FORMAT values
// (svtyper/classic.py:454-513, svtyper/singlesample.py:207-227,430-471; '%0.2f' for floats: svtyper/parsers.py:391-399).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/svtyper_hip.h"
#include "svt_fast_format.h"

namespace svt {
namespace fmt {



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
inline void put_field(std::string& s, const svt_result& r, uint8_t field, bool skipped_as_dots)
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


}  // namespace fmt
}  // namespace svt
