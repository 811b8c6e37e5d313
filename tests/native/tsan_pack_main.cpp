// ThreadSanitizer harness of the packed-evidence encoder (svt_pack.cpp is plain host C++): five calls with eight workers over a
// random three-library batch (library switches in the pair streams), the last two on a batch whose offsets are not monotone (the error path through the phase barrier).
// Built and run by tests/test_sanitizers.py with g++ -fsanitize=thread.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "svt_pack.h"
using namespace svt;
int main() {
    const uint64_t n = 20000;
    std::mt19937 rng(1);
    std::vector<uint64_t> off(n + 1, 0);
    std::vector<svt_unit> units(n);
    for (uint64_t u = 0; u < n; ++u) { off[u + 1] = off[u] + rng() % 60; units[u] = svt_unit{}; units[u].var_length = 300 + (int)(rng() % 2000); units[u].pos_delta = units[u].var_length; units[u].svtype = rng() % 3; }
    std::vector<svt_record> recs(off[n]);
    for (auto& r : recs) { std::memset(&r, 0, sizeof r); r.ospan_len = 100 + (int)(rng() % 900); r.mapq_a = 60; r.mapq_b = (rng() % 10) ? 60 : 37; r.flags = (rng() % 8) | SVT_REC_HAS_PAIR | ((rng() % 3) << SVT_REC_LIB_SHIFT); r.rs_a = (rng() % 3) ? 0 : 60; r.seq_l = (rng() % 20) ? 0 : 40; r.clip_r = (rng() % 25) ? 0 : 33; }
    std::vector<uint32_t> hist(600);
    for (size_t i = 0; i < hist.size(); ++i) hist[i] = 1 + (uint32_t)(1000.0 * std::exp(-0.5 * ((double)i - 300) * ((double)i - 300) / 6400.0));
    svt_library lib{}; lib.hist = hist.data(); lib.key_min = 50; lib.n_bins = (uint32_t)hist.size(); lib.mean = 350.37; lib.sd = 80.71;
    svt_library libs[3] = {lib, lib, lib};
    libs[1].key_min = 80; libs[1].n_bins = 500; libs[1].mean = 380.21; libs[2].key_min = 20; libs[2].sd = 40.13;
    svt_evidence_batch in{}; in.n_units = n; in.rec_offset = off.data(); in.units = units.data(); in.records = recs.data(); in.n_libs = 3; in.libs = libs; in.split_weight = 1; in.disc_weight = 1;
    PackAlloc A{[](uint64_t b) { return std::malloc(b); }, [](void* p) { std::free(p); }};
    // the ranged form (svt_genotype_packed_from_records): ranges of 2048 units handed over on the calling thread while the other
    // threads are already encoding the next range; rep 1 refuses the third range (the stop path through the later gates)
    for (int rep = 0; rep < 2; ++rep) {
        struct Seen { uint64_t units = 0, slots = 0, calls = 0; int fail_at; } seen;
        seen.fail_at = rep == 1 ? 2 : -1;
        PackSink sink;
        sink.range_units = 2048;
        sink.slots_cap = off[n] + 3 * n + 64;
        sink.ctx = &seen;
        sink.ready = [](void* ctx, const PackedArrays* a, uint64_t u0, uint64_t u1, uint64_t s0, uint64_t s1) -> int {
            Seen& s = *static_cast<Seen*>(ctx);
            if ((int)s.calls == s.fail_at) return -7;
            uint64_t x = 0;                                        // read what was handed over, as the consumer's DMA would
            for (uint64_t i = s0; i < s1; ++i) x += static_cast<const uint32_t*>(a->slots)[4 * i];
            for (uint64_t u = u0; u < u1; ++u) x += a->off[3 * u + 3] + (uint64_t)a->units[u].var_length;
            s.units += u1 - u0;
            s.slots += s1 - s0 + (x == 1 ? 0 : 0);
            ++s.calls;
            return 0;
        };
        PackedArrays out;
        int rc = encode_packed(&in, A, &out, &sink);
        std::printf("ranged %d rc %d units %llu slots %llu of %llu calls %llu\n", rep, rc, (unsigned long long)seen.units, (unsigned long long)seen.slots,
                    (unsigned long long)out.n_slots, (unsigned long long)seen.calls);
        if (rc == 0) { A.put(out.off); A.put(out.units); A.put(out.slots); }
    }
    for (int rep = 0; rep < 5; ++rep) {
        PackedArrays out;
        int rc = encode_packed(&in, A, &out);
        std::printf("rep %d rc %d slots %llu err '%s'\n", rep, rc, (unsigned long long)out.n_slots, g_err.c_str());
        if (rc == 0) { A.put(out.off); A.put(out.units); A.put(out.slots); }
        if (rep == 2) off[500] = off[499] - 1 > off[501] ? 0 : off[501] + 5;   // break monotonicity: error path through the barrier
    }
    pack_trim();
    return 0;
}
