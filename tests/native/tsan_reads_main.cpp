// ThreadSanitizer harness of the native reader (svt_reads.cpp is plain host C++): svt_bam_summarise over the units the caller
// dumped (windows + breakpoints as the ctypes layer hands them over), with one and with eight workers.  The workers share
// inflated BGZF blocks through SharedBlocks and claim runs of units from one counter: a race there would be silent, and the
// summaries must not depend on the number of workers.  Built and run by tests/test_sanitizers.py with g++ -fsanitize=thread.
//   tsan_reads <bam> <windows.bin> <breakpoints.bin> <read group>=<library index> ...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "svtyper_reads.h"
#include "svt_error.h"
// (svt_last_error lives in the HIP translation unit of the library; this program links the reader alone)
extern "C" const char* svt_last_error(void) { return svt::g_err.c_str(); }

template <class T>
static std::vector<T> slurp(const char* path)
{
    std::vector<T> v;
    if (FILE* f = std::fopen(path, "rb")) {
        std::fseek(f, 0, SEEK_END);
        const long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        v.resize((size_t)n / sizeof(T));
        if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) v.clear();
        std::fclose(f);
    }
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    svt_bam* bam = nullptr;
    if (svt_bam_open(argv[1], &bam) != 0) { std::printf("open failed: %s\n", svt_last_error()); return 3; }
    const auto win = slurp<svt_fetch_unit>(argv[2]);
    const auto bps = slurp<svt_breakpoint>(argv[3]);
    if (win.empty() || win.size() != bps.size()) return 4;
    std::vector<std::string> names;
    std::vector<int32_t> libs;
    for (int i = 4; i < argc; ++i) {
        const char* eq = std::strrchr(argv[i], '=');
        if (!eq) return 5;
        names.emplace_back((const char*)argv[i], (size_t)(eq - argv[i]));
        libs.push_back(std::atoi(eq + 1));
    }
    std::vector<const char*> name_ptrs;
    for (const auto& s : names) name_ptrs.push_back(s.c_str());
    svt_summarise_args a{};
    a.n_units = win.size();
    a.windows = win.data();
    a.breakpoints = bps.data();
    a.n_read_groups = (uint32_t)names.size();
    a.read_groups = name_ptrs.data();
    a.read_group_lib = libs.data();
    a.max_reads = 1000;
    a.count_mode = 1;
    for (int threads : {1, 8, 8}) {
        a.n_threads = threads;
        svt_summaries s{};
        const int rc = svt_bam_summarise(bam, &a, &s);
        if (rc != 0) { std::printf("threads %d rc %d: %s\n", threads, rc, svt_last_error()); return 6; }
        const uint64_t total = s.frag_offset[win.size()];
        uint64_t h = 1469598103934665603ull;                                   // FNV-1a over offsets, flags and summaries
        auto mix = [&](const void* p, size_t n) { for (size_t i = 0; i < n; ++i) h = (h ^ static_cast<const uint8_t*>(p)[i]) * 1099511628211ull; };
        mix(s.frag_offset, (win.size() + 1) * sizeof(uint64_t));
        mix(s.skipped, win.size());
        mix(s.fragments, total * sizeof(svt_fragment));
        std::printf("threads %d units %zu fragments %llu hash %016llx\n", threads, win.size(), (unsigned long long)total, (unsigned long long)h);
        svt_summaries_free(&s);
    }
    // the same units as evidence records (svt_bam_evidence: the geometry predicates in the workers)
    int32_t max_lib = 0;
    for (int32_t l : libs) max_lib = l > max_lib ? l : max_lib;
    std::vector<double> flank((size_t)max_lib + 1, 400.0);
    svt_evidence_params g{};
    g.n_libs = (uint32_t)flank.size();
    g.lib_flank = flank.data();
    g.min_aligned = 20;
    g.split_slop = 3;
    for (int threads : {1, 8, 8}) {
        a.n_threads = threads;
        svt_evidence e{};
        const int rc = svt_bam_evidence(bam, &a, &g, &e);
        if (rc != 0) { std::printf("evidence threads %d rc %d: %s\n", threads, rc, svt_last_error()); return 7; }
        const uint64_t total = e.rec_offset[win.size()];
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) { for (size_t i = 0; i < n; ++i) h = (h ^ static_cast<const uint8_t*>(p)[i]) * 1099511628211ull; };
        mix(e.rec_offset, (win.size() + 1) * sizeof(uint64_t));
        mix(e.skipped, win.size());
        mix(e.records, total * sizeof(svt_record));
        std::printf("evidence threads %d units %zu records %llu hash %016llx\n", threads, win.size(), (unsigned long long)total, (unsigned long long)h);
        svt_evidence_free(&e);
    }
    svt_bam_close(bam);
    return 0;
}
