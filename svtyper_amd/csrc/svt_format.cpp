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
#include "svt_format_fields.h"
#include "svt_host_cpus.h"

namespace {

using svt::fail;
using svt::guarded;
using svt::run_threads;
using svt::fmt::host_sample_qual;
using svt::fmt::put_field;

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
