// svt_pack.cpp -- the host encoder of packed evidence (include/svtyper_hip.h: svt_pack_evidence)
//
// Plain C++17, no HIP: it is compiled into libsvtyper_hip.so beside svtyper_hip.hip (which hands it the page-locked
// pool as allocator) and, unchanged, into the sanitizer build of the host code (csrc/Makefile: host_asan).
//
// What a producer that has to cross PCIe hands over instead of the 16-byte canonical records: per unit three sparse
// streams of 2-/4-byte entries in 16-byte slots (formats: svt_entry_formats.h).  The encoder reads every record ONCE:
// a worker takes chunks of 256 units (~0.4 MB of records), writes each unit's three streams into a small scratch
// sized for the worst case, appends the used slots to its own arena and notes the slot counts; when all chunks are
// done the counts are prefix-summed (per chunk in parallel, chunk bases serially) and the arenas are copied to their
// final place in one parallel sweep.  The record contract is checked in the same loop.  (The first version made
// two passes over the 1.6 GB of records per million units -- count, then write -- and validated the unit arrays
// serially: 113-125 ms per million units on 16 threads; this one: see DESIGN.md 3.2.)
#include <algorithm>
#include <atomic>
#include <functional>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "svt_entry_formats.h"
#include "svt_host_cpus.h"
#include "svt_host_tables.h"
#include "svt_pack.h"

namespace svt {

namespace {

struct Slot { uint32_t x, y, z, w; };   // one 16-byte slot / one canonical record, as four dwords
static_assert(sizeof(Slot) == 16 && sizeof(svt_record) == 16, "records and slots are 16 bytes");

// The streams are written as plain arrays of half-words (a slot = eight of them, little endian: dword j of a slot =
// half-word 2 j | half-word 2 j + 1 << 16), which is all the formats of svt_entry_formats.h are.
//
// seven 2-byte MAPQ-pair entries per 16-byte slot (half-words 0..6); half-word 7: bit k = entry k is the first kept
// one of its fragment, bit 8 + k = entry k is a clip candidate (candidate stream)
struct WeightStream {
    uint16_t* begin;
    uint16_t* row;
    uint32_t k = 0, bits = 0;
    explicit WeightStream(uint16_t* p) : begin(p), row(p) {}
    inline void put(const uint32_t mapq_pair, const bool first, const bool clip)
    {
        row[k] = (uint16_t)mapq_pair;
        bits |= (first ? 1u : 0u) << k | (clip ? 1u : 0u) << (8u + k);
        if (++k == 7u) {
            row[7] = (uint16_t)bits;
            row += 8;
            k = bits = 0u;
        }
    }
    // the same without a data-dependent branch: the entry is written either way and kept when keep = 1 (whether a record
    // has a reference read is close to a coin toss, which the branch predictor loses)
    inline void put_if(const uint32_t mapq_pair, const uint32_t keep, const uint32_t first)
    {
        row[k] = (uint16_t)mapq_pair;
        bits |= (keep & first) << k;
        k += keep;
        if (k == 7u) {
            row[7] = (uint16_t)bits;
            row += 8;
            k = bits = 0u;
        }
    }
    inline uint32_t finish()    // slots used
    {
        if (k) {
            for (uint32_t j = k; j < 7u; ++j) row[j] = 0;
            row[7] = (uint16_t)bits;
            row += 8;
        }
        return (uint32_t)((row - begin) >> 3);
    }
};

// pair stream: eight half-words per 16-byte slot
struct PairStream {
    uint16_t* begin;
    uint32_t n = 0;       // half-words so far
    // several libraries (svt_entry_formats.h): the library the decoder's context holds at this point of the stream, and the
    // library of the records being encoded -- the switch half-word goes out in front of the next entry that is really stored
    uint32_t cur_lib = 0, want_lib = 0;
    explicit PairStream(uint16_t* p) : begin(p) {}
    inline void sync_lib()
    {
        if (want_lib != cur_lib) {
            begin[n++] = (uint16_t)((want_lib + 1u) << 3);
            cur_lib = want_lib;
        }
    }
    inline void put(const uint32_t lo16, const uint32_t mq, const uint32_t common)
    {
        sync_lib();
        if (mq == common) {
            begin[n++] = (uint16_t)lo16;
        } else {
            begin[n] = 0;                      // no-op half-word: a wide entry starts on a 4-byte boundary
            n += n & 1u;
            begin[n++] = (uint16_t)(lo16 | kWideEntry);
            begin[n++] = (uint16_t)mq;
        }
    }
    inline uint32_t finish()
    {
        while (n & 7u) begin[n++] = 0;
        return n >> 3;
    }
};

// The most common (mapq_a, mapq_b) among the first records that would keep a pair entry (a straddle bit and two
// non-zero MAPQs); ties go to the lowest key.  Any answer is correct, a good one makes the pair stream shorter.
uint32_t vote_common_mapq(const Slot* recs, uint64_t n_vote)
{
    uint32_t common = kDefaultCommonMapq, best = 0;
    if (!recs || !n_vote) return common;
    std::vector<uint32_t> votes(65536, 0u);
    for (uint64_t i = 0; i < n_vote; ++i) {
        const Slot w = recs[i];
        if ((w.w & 7u) && (w.y & 0xffu) && (w.y & 0xff00u)) ++votes[w.y & 0xffffu];
    }
    for (uint32_t k = 0; k < 65536u; ++k)
        if (votes[k] > best) { best = votes[k]; common = k; }
    return common;
}

// what the record loop needs to know about the unit, and what it carries from record to record
struct UnitCtx {
    bool gated;                  // a DEL below the small-deletion gate of classic.py:339,383: no pair entry adds anything
    int64_t key_min, nb, vl;     // library key_min / n_bins, the unit's var_length
    uint64_t lim1, lim2;         // pair_code (svt_entry_formats.h) with the unit's constants folded: code = r when
    uint32_t code_out;           // (uint64) r < lim1, else nb + (r - vl) when (uint64)(r - vl) < lim2, else code_out = 2 nb
    uint32_t common;             // the batch's common MAPQ pair
};
struct UnitState {
    bool has_r = false, has_s = false, has_c = false;    // the fragment already has a kept entry for that tally
    uint32_t or_flags = 0, or_span = 0, lone = 0;        // the record contract, folded like the device's RecordCheck
};

#ifndef SVT_PACK_BRANCHLESS_REF
#define SVT_PACK_BRANCHLESS_REF 1
#endif

// records [r0, r1) of a unit, one at a time (also the tail and the odd groups of the vector form below)
inline void encode_records(const Slot* recs, const uint64_t r0, const uint64_t r1, const UnitCtx& c, UnitState& st,
                           PairStream& S, WeightStream& R, WeightStream& X)
{
    for (uint64_t j = r0; j < r1; ++j) {
        const Slot w = recs[j];
        const uint32_t fl = w.w;
        st.or_flags |= fl;
        st.or_span |= w.x;
        st.lone |= (fl & 7u) & (((fl >> 4) & 1u) - 1u);     // straddle bits of a record without HAS_PAIR
        if (!(fl & SVT_REC_CONTINUATION)) st.has_r = st.has_s = st.has_c = false;
        // a pair entry that could only add +0.0 is not stored: no straddle bit, a zero MAPQ (prob_mapq(0) == 0.0), a gated DEL
        if ((fl & 7u) && (w.y & 0xffu) && (w.y & 0xff00u) && !c.gated) {
            const int64_t r = (int64_t)(int32_t)w.x - c.key_min;
            const uint32_t code = (uint64_t)r < c.lim1 ? (uint32_t)r : (uint64_t)(r - c.vl) < c.lim2 ? (uint32_t)(c.nb + r - c.vl) : c.code_out;
            S.put((fl & 7u) | (code << 3), w.y & 0xffffu, c.common);
        }
        const uint32_t k_ref = w.y >> 16, k_seq = w.z & 0xffffu, k_clip = w.z >> 16;   // gated MAPQ pairs; 0 = nothing to add
#if SVT_PACK_BRANCHLESS_REF
        R.put_if(k_ref, k_ref ? 1u : 0u, st.has_r ? 0u : 1u);
        st.has_r |= k_ref != 0u;
#else
        if (k_ref) { R.put(k_ref, !st.has_r, false); st.has_r = true; }
#endif
        if (k_seq) { X.put(k_seq, !st.has_s, false); st.has_s = true; }
        if (k_clip) { X.put(k_clip, !st.has_c, true); st.has_c = true; }
    }
}

// Several libraries (svt_entry_formats.h: library switches): what the record loops know about the unit -- its constants
// against each library a record names, kept for a window of sixteen consecutive libraries (a sample's libraries are
// neighbours in the batch) and filled when a record first names one.
struct MultiUnit {
    const LibDesc* libs;
    uint32_t n_libs, common;
    bool is_del;
    int64_t vl;
    double pos_delta;
    bool bad_lib = false;            // a record names a library the batch does not have (the batch is rejected)
    uint32_t lo = 0, have = 0;       // the window [lo, lo + 16), bit k: entry k of the tables below is filled
    alignas(64) int32_t t_kmin[16], t_nb[16], t_lim1[16], t_lim2[16], t_out[16], t_gated[16];
    inline UnitCtx ctx_of(uint32_t L)
    {
        if (L >= n_libs) { bad_lib = true; L = 0; }
        const LibDesc& lib = libs[L];
        UnitCtx c;
        const int64_t nb = lib.n_bins;
        c.gated = is_del && pos_delta < lib.sd2;                  // classic.py:339,383
        c.key_min = lib.key_min;
        c.nb = nb;
        c.vl = vl;
        c.lim1 = !is_del ? (uint64_t)nb : vl < nb ? (uint64_t)(vl + nb) : (uint64_t)nb;
        c.lim2 = is_del && vl >= nb ? (uint64_t)nb : 0u;
        c.code_out = (uint32_t)(2 * nb);
        c.common = common;
        return c;
    }
    inline void fill(const uint32_t k)
    {
        const UnitCtx c = ctx_of(lo + k);
        t_kmin[k] = (int32_t)c.key_min;
        t_nb[k] = (int32_t)c.nb;
        t_lim1[k] = (int32_t)(uint32_t)c.lim1;
        t_lim2[k] = (int32_t)(uint32_t)c.lim2;
        t_out[k] = (int32_t)c.code_out;
        t_gated[k] = c.gated ? -1 : 0;
        have |= 1u << k;
    }
};

// records [r0, r1) of a unit of a batch of several libraries, as runs of one library each; the switch half-word goes out in
// front of the first entry a run really stores (PairStream::sync_lib)
inline void encode_records_runs(const Slot* recs, const uint64_t r0, const uint64_t r1, MultiUnit& M, UnitState& st,
                                PairStream& S, WeightStream& R, WeightStream& X)
{
    for (uint64_t j = r0; j < r1;) {
        const uint32_t L = SVT_REC_LIB(recs[j].w);
        uint64_t e = j + 1;
        while (e < r1 && SVT_REC_LIB(recs[e].w) == L) ++e;
        const UnitCtx c = M.ctx_of(L);
        S.want_lib = L < M.n_libs ? L : 0u;
        encode_records(recs, j, e, c, st, S, R, X);
        j = e;
    }
}

#ifndef SVT_PACK_PREFETCH
#define SVT_PACK_PREFETCH 128   // records (16 bytes each) the vector loop prefetches ahead; 0 = none
#endif
#if defined(__x86_64__)
#define SVT_PACK_AVX512 1
#include <immintrin.h>
// The same for sixteen records at a time (AVX-512 F / BW / VL; chosen at run time).  The four dwords of the records are
// transposed into four vectors; which records keep a pair entry, their codes, which carry a reference read or a split
// candidate come out of a dozen vector instructions instead of sixteen times a dozen branches; the entries themselves
// are then emitted in record order: runs of one-half-word pair entries by a compressing store, the few wide ones and
// the weight entries from the set bits of the masks.  A group that holds a continuation record (a fragment with a
// second record: the first-of-fragment bits then depend on the records before it) is left to encode_records.
// one-half-word pair entries `ent` of the lanes in `run`, of libraries `libv`: compressed to the front; a switch half-word goes
// in front of every entry whose library differs from the entry before it (the first one: from the stream's current library).
// Switches and entries interleaved lane by lane (s0 e0 s1 e1 ...), the lanes that exist compressed once more and stored.
__attribute__((target("avx512f,avx512bw,avx512vl,bmi2")))
inline void emit_mixed_run(PairStream& S, const __m512i ent, const __m512i libv, const __mmask16 run, const uint32_t n_libs)
{
    const unsigned k = (unsigned)__builtin_popcount(run);
    const __mmask16 kmask = (__mmask16)((1u << k) - 1u);
    const __m512i ec = _mm512_maskz_compress_epi32(run, ent), lc = _mm512_maskz_compress_epi32(run, libv);
    const __m512i prev = _mm512_alignr_epi32(lc, _mm512_set1_epi32((int32_t)S.cur_lib), 15);    // lane i: library of entry i - 1
    const __mmask16 changed = _mm512_mask_cmpneq_epi32_mask(kmask, lc, prev);
    const __m512i sw = _mm512_slli_epi32(_mm512_add_epi32(lc, _mm512_set1_epi32(1)), 3);
    const __m512i ia = _mm512_setr_epi32(0, 16, 1, 17, 2, 18, 3, 19, 4, 20, 5, 21, 6, 22, 7, 23);
    const __m512i ib = _mm512_setr_epi32(8, 24, 9, 25, 10, 26, 11, 27, 12, 28, 13, 29, 14, 30, 15, 31);
    const __mmask16 ma = (__mmask16)(_pdep_u32(changed & 0xffu, 0x5555u) | _pdep_u32(kmask & 0xffu, 0xAAAAu));
    const __mmask16 mb = (__mmask16)(_pdep_u32((unsigned)changed >> 8, 0x5555u) | _pdep_u32((unsigned)kmask >> 8, 0xAAAAu));
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(S.begin + S.n),
                        _mm512_cvtepi32_epi16(_mm512_maskz_compress_epi32(ma, _mm512_permutex2var_epi32(sw, ia, ec))));
    S.n += (uint32_t)__builtin_popcount(ma);
    if (mb) {
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(S.begin + S.n),
                            _mm512_cvtepi32_epi16(_mm512_maskz_compress_epi32(mb, _mm512_permutex2var_epi32(sw, ib, ec))));
        S.n += (uint32_t)__builtin_popcount(mb);
    }
    const uint32_t last_lib = (uint32_t)_mm_cvtsi128_si32(_mm512_castsi512_si128(_mm512_maskz_compress_epi32((__mmask16)(1u << (k - 1u)), lc)));
    S.cur_lib = S.want_lib = last_lib < n_libs ? last_lib : 0u;
}

// MULTI (a batch of several libraries; c is not used, M is the unit): the unit's constants come per record from M's
// window tables by a lane permutation; a group whose kept pair entries are all of one library is emitted as before (behind
// a switch if that library is not the stream's current one), a mixed group entry by entry from the vector's lanes -- the
// arithmetic stays in vectors either way.  A group that names libraries more than sixteen apart goes record by record.
template <bool MULTI>
__attribute__((target("avx512f,avx512bw,avx512vl,bmi2")))
inline void encode_records_avx512(const Slot* recs, const uint64_t r0, const uint64_t r1, const UnitCtx& c, MultiUnit* M, UnitState& st,
                                  PairStream& S, WeightStream& R, WeightStream& X)
{
    // the gated unit keeps no pair entry at all; negative codes cannot happen for it either way
    const __m512i idx_lo = _mm512_setr_epi32(0, 4, 8, 12, 16, 20, 24, 28, 0, 0, 0, 0, 0, 0, 0, 0);      // dword 0 of records 0..7 of (A, B)
    const __m512i idx_hi = _mm512_setr_epi32(0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 8, 12, 16, 20, 24, 28);      // ... into lanes 8..15
    const __m512i one = _mm512_set1_epi32(1);
    const __m256i iota16 = _mm256_setr_epi16(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    __m512i acc_flags = _mm512_setzero_si512(), acc_span = _mm512_setzero_si512(), acc_lone = _mm512_setzero_si512();
    const __m512i v_vl = _mm512_set1_epi32((int32_t)(MULTI ? M->vl : c.vl));
    __m512i v_kmin = _mm512_set1_epi32((int32_t)c.key_min), v_nb = _mm512_set1_epi32((int32_t)c.nb), v_out = _mm512_set1_epi32((int32_t)c.code_out);
    __m512i v_lim1 = _mm512_set1_epi32((int32_t)(uint32_t)c.lim1), v_lim2 = _mm512_set1_epi32((int32_t)(uint32_t)c.lim2);
    const __m512i v_common = _mm512_set1_epi32((int32_t)(MULTI ? M->common : c.common));
    uint64_t j = r0;
    for (; j < r1; j += 16) {
        const unsigned n_here = (unsigned)std::min<uint64_t>(16, r1 - j);   // < 16: the unit's last records, the lanes behind them read as all-zero records
#if SVT_PACK_PREFETCH
        // the records are a pure stream: ask for the lines a few groups ahead (sixteen workers share the memory system, and
        // what the hardware prefetcher brings on its own arrives late under that load)
        _mm_prefetch(reinterpret_cast<const char*>(recs + j + SVT_PACK_PREFETCH), _MM_HINT_T0);
        _mm_prefetch(reinterpret_cast<const char*>(recs + j + SVT_PACK_PREFETCH + 4), _MM_HINT_T0);
        _mm_prefetch(reinterpret_cast<const char*>(recs + j + SVT_PACK_PREFETCH + 8), _MM_HINT_T0);
        _mm_prefetch(reinterpret_cast<const char*>(recs + j + SVT_PACK_PREFETCH + 12), _MM_HINT_T0);
#endif
        __m512i a, b, d, e;
        if (n_here == 16u) {
            a = _mm512_loadu_si512(recs + j);
            b = _mm512_loadu_si512(recs + j + 4);
            d = _mm512_loadu_si512(recs + j + 8);
            e = _mm512_loadu_si512(recs + j + 12);
        } else {
            // (a masked load does not touch what it masks out: nothing behind the unit's last record is read.)  An all-zero
            // record keeps no entry of any kind and sets no contract bit, so the group's arithmetic below needs no mask
            const uint64_t dwords = ((uint64_t)1 << (4u * n_here)) - 1u;      // four dwords per record
            a = _mm512_maskz_loadu_epi32((__mmask16)(dwords & 0xffffu), recs + j);
            b = _mm512_maskz_loadu_epi32((__mmask16)((dwords >> 16) & 0xffffu), recs + j + 4);
            d = _mm512_maskz_loadu_epi32((__mmask16)((dwords >> 32) & 0xffffu), recs + j + 8);
            e = _mm512_maskz_loadu_epi32((__mmask16)((dwords >> 48) & 0xffffu), recs + j + 12);
        }
        // dword K of the sixteen records (a lambda would not inherit this function's target attribute)
#define SVT_FIELD(K)                                                                                                            \
    _mm512_mask_blend_epi32(0xFF00, _mm512_permutex2var_epi32(a, _mm512_add_epi32(idx_lo, _mm512_set1_epi32(K)), b),             \
                            _mm512_permutex2var_epi32(d, _mm512_add_epi32(idx_hi, _mm512_set1_epi32(K)), e))
        const __m512i fw = SVT_FIELD(3);
        if (_mm512_test_epi32_mask(fw, _mm512_set1_epi32((int32_t)SVT_REC_CONTINUATION))) {   // (rare) a fragment goes on: record by record
            if (MULTI) encode_records_runs(recs, j, j + n_here, *M, st, S, R, X);
            else encode_records(recs, j, j + n_here, c, st, S, R, X);
            continue;
        }
        __m512i libv = _mm512_setzero_si512();
        __mmask16 gated_lanes = 0;
        if (MULTI) {
            // every record's library as an index into the unit's window of sixteen; the window moves to the group's libraries
            // when a record falls outside it, its entries are filled when first named
            const __mmask16 valid = (__mmask16)((1u << n_here) - 1u);
            libv = _mm512_and_si512(_mm512_srli_epi32(fw, SVT_REC_LIB_SHIFT), _mm512_set1_epi32(0xff));
            __m512i idx = _mm512_sub_epi32(libv, _mm512_set1_epi32((int32_t)M->lo));
            if (_mm512_mask_cmpge_epu32_mask(valid, idx, _mm512_set1_epi32(16))) {
                const uint32_t gmin = _mm512_mask_reduce_min_epu32(valid, libv), gmax = _mm512_mask_reduce_max_epu32(valid, libv);
                if (gmax - gmin >= 16u) {
                    encode_records_runs(recs, j, j + n_here, *M, st, S, R, X);
                    continue;
                }
                M->lo = gmin;
                M->have = 0;
                idx = _mm512_sub_epi32(libv, _mm512_set1_epi32((int32_t)gmin));
            }
            const __m512i bit = _mm512_sllv_epi32(one, idx);
            if (_mm512_mask_testn_epi32_mask(valid, bit, _mm512_set1_epi32((int32_t)M->have))) {     // a library named for the first time
                const uint32_t present = (uint32_t)_mm512_mask_reduce_or_epi32(valid, bit);
                for (uint32_t miss = present & ~M->have; miss; miss &= miss - 1u) M->fill((uint32_t)__builtin_ctz(miss));
            }
            v_kmin = _mm512_permutexvar_epi32(idx, _mm512_load_si512(M->t_kmin));
            v_nb = _mm512_permutexvar_epi32(idx, _mm512_load_si512(M->t_nb));
            v_lim1 = _mm512_permutexvar_epi32(idx, _mm512_load_si512(M->t_lim1));
            v_lim2 = _mm512_permutexvar_epi32(idx, _mm512_load_si512(M->t_lim2));
            v_out = _mm512_permutexvar_epi32(idx, _mm512_load_si512(M->t_out));
            const __m512i g = _mm512_permutexvar_epi32(idx, _mm512_load_si512(M->t_gated));
            gated_lanes = _mm512_test_epi32_mask(g, g);
        }
        const __m512i fx = SVT_FIELD(0), fy = SVT_FIELD(1), fz = SVT_FIELD(2);
#undef SVT_FIELD
        acc_flags = _mm512_or_si512(acc_flags, fw);
        acc_span = _mm512_or_si512(acc_span, fx);
        // straddle bits of a record without HAS_PAIR: (fl & 7) & (((fl >> 4) & 1) - 1)
        acc_lone = _mm512_or_si512(acc_lone, _mm512_and_si512(_mm512_and_si512(fw, _mm512_set1_epi32(7)),
                                                              _mm512_sub_epi32(_mm512_and_si512(_mm512_srli_epi32(fw, 4), one), one)));
        // ---- pair entries
        __mmask16 keep = 0;
        if (MULTI || !c.gated)
            keep = _mm512_test_epi32_mask(fw, _mm512_set1_epi32(7)) & _mm512_test_epi32_mask(fy, _mm512_set1_epi32(0xff)) &
                   _mm512_test_epi32_mask(fy, _mm512_set1_epi32(0xff00));
        if (MULTI) keep &= (__mmask16)~gated_lanes;
        bool mixed = false;      // several libraries among the kept entries of this group
        if (MULTI && keep) {
            const uint32_t first_lib = (uint32_t)_mm_cvtsi128_si32(_mm512_castsi512_si128(_mm512_maskz_compress_epi32(keep, libv)));
            mixed = _mm512_mask_cmpneq_epi32_mask(keep, libv, _mm512_set1_epi32((int32_t)first_lib)) != 0;
            S.want_lib = first_lib < M->n_libs ? first_lib : 0u;
        }
        if (keep) {
            if (!mixed) S.sync_lib();
            // r = ospan_len - key_min as the 32-bit value it is under the format's limits (|key_min| <= 2^29, ospan_len >= 0
            // or rejected): a negative r is a huge unsigned number and fails both range tests, like the 64-bit form
            const __m512i r = _mm512_sub_epi32(fx, v_kmin), r2 = _mm512_sub_epi32(r, v_vl);
            const __mmask16 in1 = _mm512_cmplt_epu32_mask(r, v_lim1), in2 = _mm512_cmplt_epu32_mask(r2, v_lim2);
            __m512i code = _mm512_mask_blend_epi32(in2, v_out, _mm512_add_epi32(v_nb, r2));
            code = _mm512_mask_blend_epi32(in1, code, r);
            const __m512i mq = _mm512_and_si512(fy, _mm512_set1_epi32(0xffff));
            const __m512i ent = _mm512_or_si512(_mm512_and_si512(fw, _mm512_set1_epi32(7)), _mm512_slli_epi32(code, 3));
            __mmask16 wide = keep & _mm512_cmpneq_epi32_mask(mq, v_common);
            if (MULTI && mixed) {
                // one-half-word entries in runs between the (few) wide ones, each run with its switches (emit_mixed_run)
                alignas(64) uint32_t e32[16], m32[16], l32[16];
                if (wide) {
                    _mm512_store_si512(e32, ent);
                    _mm512_store_si512(m32, mq);
                    _mm512_store_si512(l32, libv);
                }
                unsigned from = 0;
                while (true) {
                    const unsigned i = wide ? (unsigned)__builtin_ctz(wide) : 16u;
                    const __mmask16 run = (__mmask16)(keep & ((1u << i) - 1u) & ~((1u << from) - 1u));
                    if (run) emit_mixed_run(S, ent, libv, run, M->n_libs);
                    if (i == 16u) break;
                    S.want_lib = l32[i] < M->n_libs ? l32[i] : 0u;
                    S.put(e32[i], m32[i], M->common);
                    from = i + 1u;
                    wide = (__mmask16)(wide & (wide - 1u));
                }
            } else if (!wide) {
                // all of them one half-word: compress the kept entries and store them as sixteen half-words (the
                // scratch has room; what lies behind the kept ones is overwritten by whatever comes next)
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(S.begin + S.n), _mm512_cvtepi32_epi16(_mm512_maskz_compress_epi32(keep, ent)));
                S.n += (uint32_t)__builtin_popcount(keep);
            } else {
                alignas(64) uint32_t e32[16], m32[16];
                _mm512_store_si512(e32, ent);
                _mm512_store_si512(m32, mq);
                unsigned from = 0;
                while (true) {
                    const unsigned i = wide ? (unsigned)__builtin_ctz(wide) : 16u;
                    const __mmask16 run = (__mmask16)(keep & ((1u << i) - 1u) & ~((1u << from) - 1u));   // one-half-word entries in front of lane i
                    if (run) {
                        _mm256_storeu_si256(reinterpret_cast<__m256i*>(S.begin + S.n), _mm512_cvtepi32_epi16(_mm512_maskz_compress_epi32(run, ent)));
                        S.n += (uint32_t)__builtin_popcount(run);
                    }
                    if (i == 16u) break;
                    S.begin[S.n] = 0;                      // the no-op half-word in front of a wide entry at an odd half-word
                    S.n += S.n & 1u;
                    S.begin[S.n++] = (uint16_t)(e32[i] | kWideEntry);
                    S.begin[S.n++] = (uint16_t)m32[i];
                    from = i + 1u;
                    wide = (__mmask16)(wide & (wide - 1u));
                }
            }
        }
        // ---- reference reads (every record of the group starts a fragment: each kept entry is its fragment's first)
        const __m512i kref = _mm512_srli_epi32(fy, 16);
        __mmask16 nz = _mm512_test_epi32_mask(kref, kref);
        if (nz) {
            // the kept entries, compressed to the front; a row takes what it has room for (seven entries + the bits half-word),
            // the rest moves down by a half-word permutation.  Each store writes sixteen half-words: what lies behind the
            // entries just placed is overwritten by the next store, the row's bits half-word or finish() (the scratch has the room)
            __m256i v = _mm512_cvtepi32_epi16(_mm512_maskz_compress_epi32(nz, kref));
            uint32_t cnt = (uint32_t)__builtin_popcount(nz);
            for (;;) {
                const uint32_t take = std::min(cnt, 7u - R.k);
                _mm256_storeu_si256(reinterpret_cast<__m256i*>(R.row + R.k), v);
                R.bits |= ((1u << take) - 1u) << R.k;
                R.k += take;
                cnt -= take;
                if (R.k == 7u) {
                    R.row[7] = (uint16_t)R.bits;
                    R.row += 8;
                    R.k = R.bits = 0u;
                }
                if (!cnt) break;
                v = _mm256_permutexvar_epi16(_mm256_add_epi16(iota16, _mm256_set1_epi16((short)take)), v);
            }
        }
        // ---- split / clip candidates (few): in record order, the split candidate of a record before its clip candidate
        const __mmask16 any_x = _mm512_test_epi32_mask(fz, fz);
        if (any_x) {
            alignas(64) uint32_t z32[16];
            _mm512_store_si512(z32, fz);
            for (unsigned m = any_x; m; m &= m - 1u) {
                const uint32_t z = z32[__builtin_ctz(m)];
                if (z & 0xffffu) X.put(z & 0xffffu, true, false);
                if (z >> 16) X.put(z >> 16, true, true);
            }
        }
        // what a continuation record right behind this group would see of its fragment (the group's last record)
        const uint32_t last_y = recs[j + n_here - 1u].y, last_z = recs[j + n_here - 1u].z;
        st.has_r = (last_y >> 16) != 0u;
        st.has_s = (last_z & 0xffffu) != 0u;
        st.has_c = (last_z >> 16) != 0u;
    }
    st.or_flags |= (uint32_t)_mm512_reduce_or_epi32(acc_flags);
    st.or_span |= (uint32_t)_mm512_reduce_or_epi32(acc_span);
    st.lone |= (uint32_t)_mm512_reduce_or_epi32(acc_lone);
}

// memcpy whose stores bypass the caches: the destination is the page-locked output array, which the CPU never reads
// again (the next reader is the DMA engine) -- ordinary stores would first fetch every destination line for ownership.
__attribute__((target("avx512f")))
inline void copy_streaming(void* dst, const void* src, size_t bytes)
{
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    const size_t head = std::min(bytes, (size_t)((64u - (reinterpret_cast<uintptr_t>(d) & 63u)) & 63u));
    std::memcpy(d, s, head);
    d += head; s += head; bytes -= head;
    for (; bytes >= 64; d += 64, s += 64, bytes -= 64) _mm512_stream_si512(reinterpret_cast<__m512i*>(d), _mm512_loadu_si512(s));
    std::memcpy(d, s, bytes);
    _mm_sfence();
}

inline bool cpu_has_avx512()
{
    static const bool yes = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
                            __builtin_cpu_supports("bmi2");
    return yes;
}
#endif

constexpr uint64_t kChunkUnits = 256;
static_assert(SVT_REC_CONTINUATION == (1u << 3) && SVT_REC_HAS_PAIR == (1u << 4), "bit positions used by the encoder's loop");

struct ChunkOut {          // where a chunk's slots wait for the final copy
    const Slot* src = nullptr; // in the arena of the worker that encoded it: the memory never moves (segments)
    uint64_t n_slots = 0;
    uint64_t base = 0;     // first slot in the final array
};

// A worker's arena: plain memory that is neither zero-filled when it grows nor handed back between calls.  Fresh pages
// cost a page fault each and concurrent faults of one process serialise in the kernel: with arenas allocated per call
// sixteen threads ran at half the per-thread speed of one.
// It is a list of SEGMENTS that never move: a finished chunk is copied to the final array by whichever thread gets to it,
// possibly while its owner is already appending the next chunks -- growing by realloc would pull the memory away under that
// reader.  A chunk is contiguous: when the current segment cannot take its next unit, what the chunk has so far moves on to
// a larger segment (only the unfinished chunk, which nobody else can see yet).
struct Arena {
    struct Seg { Slot* p; size_t size, cap; };
    static constexpr size_t kSegSlots = size_t(1) << 18;   // 4 MB
    std::vector<Seg> segs;
    size_t cur = 0, chunk_at = 0;
    size_t capacity() const { size_t c = 0; for (const Seg& g : segs) c += g.cap; return c; }
    void add(const size_t cap)
    {
        Slot* q = static_cast<Slot*>(std::malloc(cap * sizeof(Slot)));
        if (!q) throw std::bad_alloc();
        segs.push_back(Seg{q, 0, cap});
    }
    void reset()
    {
        for (Seg& g : segs) g.size = 0;
        cur = chunk_at = 0;
    }
    void reserve_first(const size_t want) { if (segs.empty()) add(std::max(want, kSegSlots)); }
    void begin_chunk()
    {
        if (segs.empty()) add(kSegSlots);
        chunk_at = segs[cur].size;
    }
    Slot* append(const size_t n)   // room for n more slots of the chunk being written
    {
        if (segs[cur].size + n > segs[cur].cap) {
            const size_t have = segs[cur].size - chunk_at;
            size_t nxt = cur + 1;
            while (nxt < segs.size() && segs[nxt].cap < have + n) ++nxt;     // (segments too small for this chunk stay unused this call)
            if (nxt == segs.size()) add(std::max(kSegSlots, 2 * (have + n)));
            if (have) std::memcpy(segs[nxt].p, segs[cur].p + chunk_at, have * sizeof(Slot));
            segs[cur].size = chunk_at;
            cur = nxt;
            segs[cur].size = have;
            chunk_at = 0;
        }
        Slot* w = segs[cur].p + segs[cur].size;
        segs[cur].size += n;
        return w;
    }
    const Slot* chunk_begin() const { return segs[cur].p + chunk_at; }
    size_t chunk_size() const { return segs[cur].size - chunk_at; }
    void release()
    {
        for (Seg& g : segs) std::free(g.p);
        segs.clear();
    }
};
struct ArenaPool {       // arenas wait here for the next svt_pack_evidence call (released by svt_pack_trim)
    std::mutex lock;
    std::vector<Arena> idle;
    Arena get()
    {
        std::lock_guard<std::mutex> g(lock);
        if (idle.empty()) return Arena{};
        size_t best = 0;
        for (size_t i = 1; i < idle.size(); ++i)
            if (idle[i].capacity() > idle[best].capacity()) best = i;
        Arena a = std::move(idle[best]);
        idle.erase(idle.begin() + (long)best);
        a.reset();
        return a;
    }
    void put(Arena a)
    {
        if (a.segs.empty()) return;
        std::lock_guard<std::mutex> g(lock);
        if (idle.size() >= 64) { a.release(); return; }
        idle.push_back(std::move(a));
    }
    void trim()
    {
        std::lock_guard<std::mutex> g(lock);
        for (Arena& a : idle) a.release();
        idle.clear();
    }
};
ArenaPool g_arenas;

struct Worker {
    Arena arena;                   // the slots of this worker's chunks, chunk after chunk
    double ms = 0.0;               // SVT_TRACE: how long this worker ran
    ~Worker() { g_arenas.put(std::move(arena)); }
    std::vector<uint16_t> scratch; // one unit's three streams at worst-case size, as half-words
    std::atomic<uint32_t> bad{0};  // record-contract bits (kErr*); read by other threads while the owner still encodes (streamed form)
    std::atomic<int> unit_error{0};// first unit-array violation (1-based code below), 0 = none
};

enum UnitError { kUnitOk = 0, kUnitOffsets, kUnitTooLong, kUnitSvtype, kUnitReserved, kUnitVarLength, kUnitNegativeDel };

}  // namespace

void pack_trim() { g_arenas.trim(); }

namespace {
// The CPUs this process may use, grouped by the L3 cache they share (one group per CCD on an EPYC), with the socket
// each group sits on and its number of physical cores.  The encoder counts the cores next to the records (their NUMA
// node, else the calling thread's socket) to decide how many workers a short call starts.  Where the workers then run is
// the scheduler's business by default: with sixteen workers pinning them to the records' socket was worth 25 % (the remote
// half took 42 ms where the local half took 32), but with one worker per core every pinned form lost to the scheduler --
// SVT_PACK_SPREAD=1 keeps a worker inside one L3 group (SMT siblings end up sharing cores: 20-27 ms instead of 13-18),
// =2 on one CPU of it (same best time, 35-90 ms whenever another tenant's thread sits on that CPU).
struct L3Group {
    cpu_set_t cpus;
    long package;   // socket
    long node;      // NUMA node of the group's first CPU
    unsigned cores; // physical cores among `cpus` (distinct core ids)
    std::vector<long> core_ids;
};
const std::vector<L3Group>& l3_groups()
{
    static const std::vector<L3Group> groups = [] {
        std::vector<L3Group> out;
        std::vector<long> ids;
        cpu_set_t mine;
        if (sched_getaffinity(0, sizeof mine, &mine) != 0) return out;
        auto read_long = [](const char* fmt, int cpu) {
            char path[160];
            std::snprintf(path, sizeof path, fmt, cpu);
            long v = -1;
            if (FILE* f = std::fopen(path, "r")) {
                if (std::fscanf(f, "%ld", &v) != 1) v = -1;
                std::fclose(f);
            }
            return v;
        };
        for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) {
            if (!CPU_ISSET(cpu, &mine)) continue;
            const long id = read_long("/sys/devices/system/cpu/cpu%d/cache/index3/id", cpu);
            const long pkg = read_long("/sys/devices/system/cpu/cpu%d/topology/physical_package_id", cpu);
            if (id < 0 || pkg < 0) return std::vector<L3Group>();   // topology not readable: no placement
            const long core = read_long("/sys/devices/system/cpu/cpu%d/topology/core_id", cpu);
            const long key = pkg * 100000 + id;
            size_t g = 0;
            while (g < ids.size() && ids[g] != key) ++g;
            if (g == ids.size()) {
                ids.push_back(key);
                L3Group ng;
                CPU_ZERO(&ng.cpus);
                ng.package = pkg;
                ng.node = -2;   // filled below
                ng.cores = 0;
                out.push_back(ng);
            }
            CPU_SET(cpu, &out[g].cpus);
            if (std::find(out[g].core_ids.begin(), out[g].core_ids.end(), core) == out[g].core_ids.end()) {
                out[g].core_ids.push_back(core);
                ++out[g].cores;
            }
        }
        return out;
    }();
    return groups;
}
// the NUMA node most of [p, p + bytes) lives on, asked from the kernel for a few sample pages (-1: unknown)
long node_of_memory(const void* p, uint64_t bytes)
{
    if (!p || bytes < (1u << 20)) return -1;
    constexpr int kSamples = 16;
    void* pages[kSamples];
    int status[kSamples];
    for (int i = 0; i < kSamples; ++i) {
        const uint64_t at = (bytes / kSamples) * (uint64_t)i;
        pages[i] = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(p) + at) & ~uintptr_t(4095));
        status[i] = -1;
    }
    if (syscall(SYS_move_pages, 0, (unsigned long)kSamples, pages, nullptr, status, 0) != 0) return -1;
    int votes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < kSamples; ++i)
        if (status[i] >= 0 && status[i] < 8) ++votes[status[i]];
    int best = 0;
    for (int k = 1; k < 8; ++k)
        if (votes[k] > votes[best]) best = k;
    return votes[best] ? best : -1;
}
long node_of_cpu(int cpu)
{
    for (int node = 0; node < 8; ++node) {
        char path[160];
        std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/node%d", cpu, node);
        if (access(path, F_OK) == 0) return node;
    }
    return -1;
}
long package_of_cpu(int cpu)
{
    for (const L3Group& g : l3_groups())
        if (cpu >= 0 && cpu < CPU_SETSIZE && CPU_ISSET(cpu, &g.cpus)) return g.package;
    return -1;
}
}  // namespace

std::string record_error_text(uint32_t err_bits)
{
    std::string m = "invalid evidence records:";
    if (err_bits & kErrStraddleNoPair) m += " straddle bits without HAS_PAIR;";
    if (err_bits & kErrLibIndex) m += " lib index >= n_libs;";
    if (err_bits & kErrReservedBits) m += " reserved/undefined bits set;";
    if (err_bits & kErrNegativeSpan) m += " negative ospan_len;";
    return m;
}

namespace {
// The encoder's threads, started once per call (sixty-four threads cost ~1.5 ms to start) and taken through the ranges of the
// batch: for range r every thread runs first(r, t) (t = 0: the caller); when all of them are through, the caller runs
// between(r) alone; if that says SVT_OK every thread runs second(r, t); when second(r, .) has returned everywhere the caller
// runs after(r) alone -- the hand-over of the range -- and follows the others, who go on to first(r + 1, .) in the meantime.  A thread that cannot be started is simply missing: the phases hand out their work
// through atomic counters.  Waiting threads back off to short sleeps (between() may take a page-locked allocation's tens
// of ms the first time).  An exception in any thread is rethrown on the caller after the join, like run_threads does.
template <typename First, typename Between, typename Second, typename After>
int run_ranged_phases(unsigned nt, uint64_t n_ranges, First&& first, Between&& between, Second&& second, After&& after)
{
    struct Gate {
        std::atomic<unsigned> arrived{0}, finished{0};
        std::atomic<int> go{0};              // 1: second phase, -1: stop
    };
    std::vector<Gate> gates(n_ranges);
    std::atomic<unsigned> expected{~0u};
    std::exception_ptr thrown;
    std::mutex lock;
    int rc = SVT_OK;
    auto guarded_call = [&](auto&& f) {
        try {
            f();
        } catch (...) {
            std::lock_guard<std::mutex> g(lock);
            if (!thrown) thrown = std::current_exception();
        }
    };
    auto failed = [&]() -> bool {
        std::lock_guard<std::mutex> g(lock);
        return (bool)thrown;
    };
    auto wait_until = [](auto&& done) {
        for (unsigned spins = 0; !done(); ++spins) {
            if (spins < 4096) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            } else {
                std::this_thread::sleep_for(std::chrono::microseconds(30));
            }
        }
    };
    auto body = [&](unsigned t) {
        for (uint64_t r = 0; r < n_ranges; ++r) {
            Gate& G = gates[r];
            guarded_call([&] { first(r, t); });
            G.arrived.fetch_add(1, std::memory_order_acq_rel);
            if (t == 0) {
                wait_until([&] { return G.arrived.load(std::memory_order_acquire) == expected.load(std::memory_order_acquire); });
                if (!failed()) guarded_call([&] { rc = between(r); });
                G.go.store(!failed() && rc == SVT_OK ? 1 : -1, std::memory_order_release);
            } else {
                wait_until([&] { return G.go.load(std::memory_order_acquire) != 0; });
            }
            if (G.go.load(std::memory_order_acquire) != 1) return;
            guarded_call([&] { second(r, t); });
            G.finished.fetch_add(1, std::memory_order_acq_rel);
            // (second(r, .) of one thread reads what first(r, .) of the others left in THEIR arenas while those may already be
            // appending range r + 1: the arenas are segments that never move)
            if (t == 0) {
                wait_until([&] { return G.finished.load(std::memory_order_acquire) == expected.load(std::memory_order_acquire); });
                if (!failed()) guarded_call([&] { rc = after(r); });
                if (failed() || rc != SVT_OK) {
                    // (the others are in first(r + 1, .) or waiting at its gate: stop them there)
                    for (uint64_t q = r + 1; q < n_ranges; ++q) gates[q].go.store(-1, std::memory_order_release);
                    return;
                }
            }
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(nt ? nt - 1 : 0);
    unsigned started = 1;
    for (; started < nt; ++started) {
        try {
            pool.emplace_back(body, started);
        } catch (const std::system_error&) {
            break;
        }
    }
    expected.store(started, std::memory_order_release);
    body(0);
    for (auto& th : pool) th.join();
    if (thrown) std::rethrow_exception(thrown);
    return rc;
}
}  // namespace

int encode_packed(const svt_evidence_batch* in, const PackAlloc& A, PackedArrays* out, const PackSink* sink)
{
    if (!in || !out) return fail(SVT_ERR_INVALID, "null argument");
    *out = PackedArrays{};
    const uint64_t n = in->n_units;
    if (n >= 0x55555550ull) return fail(SVT_ERR_INVALID, "too many units in one batch");
    if (in->n_libs == 0 || in->n_libs > 65536 || !in->libs) return fail(SVT_ERR_INVALID, "n_libs must be 1..65536");
    if (in->n_libs > 256) return fail(SVT_ERR_UNSUPPORTED, "packed evidence names a library with eight bits: a batch of more than 256 libraries stays canonical");
    if (n && (!in->rec_offset || !in->units)) return fail(SVT_ERR_INVALID, "null unit arrays");
    if (n && in->rec_offset[0] != 0) return fail(SVT_ERR_INVALID, "rec_offset[0] must be 0");
    if (!(in->split_weight >= 0.0) || !(in->disc_weight >= 0.0) || !std::isfinite(in->split_weight) || !std::isfinite(in->disc_weight))
        return fail(SVT_ERR_INVALID, "weights must be finite and >= 0");
    // (rec_offset[n] is only trusted once the offsets below it have been seen monotone: a worker never reads a record
    // beyond rec_offset[u + 1] of a unit whose own range it has checked against its neighbours)
    const bool trace = std::getenv("SVT_TRACE") != nullptr;   // stage times on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[svt] pack: %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    HostTables T;
    SVT_TRY(build_tables(in, 0, T));
    for (const LibDesc& L : T.libs)
        if (L.n_bins > kMaxShortBins) return fail(SVT_ERR_UNSUPPORTED, "histogram too wide for the packed pair entries");
    if (!T.fast_geometry) return fail(SVT_ERR_UNSUPPORTED, "library geometry outside the packed format's range");
    const LibDesc* libs = T.libs.data();
    const uint32_t n_libs = in->n_libs;
    const bool multi = n_libs > 1;     // several libraries: library switches in the pair stream (svt_entry_formats.h)
    const Slot* recs = reinterpret_cast<const Slot*>(in->records);
    const uint64_t n_rec_claimed = n ? in->rec_offset[n] : 0;
    if (n_rec_claimed && !in->records) return fail(SVT_ERR_INVALID, "null records");

    struct Release {
        const PackAlloc& A;
        PackedArrays* p;
        const PackSink* sink;
        bool armed = true;
        ~Release()
        {
            if (!armed) return;
            if (sink && sink->drain) sink->drain(sink->ctx);   // (ranges already handed over may still be on their way out of these arrays)
            A.put(p->off); A.put(p->units); A.put(p->slots);
            *p = PackedArrays{};
        }
    } release{A, out, sink};
    out->off = static_cast<uint32_t*>(A.get((3 * n + 1) * sizeof(uint32_t)));
    out->units = static_cast<svt_unit*>(A.get(std::max<uint64_t>(n, 1) * sizeof(svt_unit)));
    if (!out->off || !out->units) return fail(SVT_ERR_NOMEM, "out of host memory");
    uint32_t* off = out->off;
    off[0] = 0u;

    // the batch's most common MAPQ pair, voted on the first records
    const uint32_t common = vote_common_mapq(recs, std::min<uint64_t>(n_rec_claimed, kVoteRecords));
    out->common = common;
    mark("tables + allocations");
    const uint64_t n_chunks = (n + kChunkUnits - 1) / kChunkUnits;
    // ranges of whole chunks: one for the plain call, several when a sink takes the evidence over as it is produced
    const uint64_t chunks_per_range = sink && sink->range_units ? std::max<uint64_t>(1, (sink->range_units + kChunkUnits - 1) / kChunkUnits)
                                                                 : std::max<uint64_t>(n_chunks, 1);
    const uint64_t n_ranges = std::max<uint64_t>(1, (n_chunks + chunks_per_range - 1) / chunks_per_range);
    if (sink) {   // the consumer copies ranges out of the slot array while later ones are written: it cannot move
        out->slots = A.get(std::max<uint64_t>(sink->slots_cap, 1) * 16);
        if (!out->slots) return fail(SVT_ERR_NOMEM, "out of host memory");
    }
    const bool read_only_probe = std::getenv("SVT_PACK_PROBE") != nullptr;
    bool use_avx512 = false;
#if SVT_PACK_AVX512
    use_avx512 = cpu_has_avx512() && std::getenv("SVT_PACK_SCALAR") == nullptr;   // (SVT_PACK_SCALAR: tests compare the two forms)
#endif
    // ---- where the workers run: the L3 groups next to the records (on their NUMA node, else on the caller's socket)
    const char* spread_env = std::getenv("SVT_PACK_SPREAD");
    const int spread = n_chunks > 1 && spread_env ? std::atoi(spread_env) : 0;   // 0: placement is the scheduler's business, 1: a worker stays in its L3 group, 2: on one CPU of it
    std::vector<const L3Group*> home;
    if (n_chunks > 1) {
        const long node = node_of_memory(in->records, n_rec_claimed * 16);
        if (node >= 0) {
            for (const L3Group& g : l3_groups()) {
                int first = -1;
                for (int cpu = 0; cpu < CPU_SETSIZE && first < 0; ++cpu)
                    if (CPU_ISSET(cpu, &g.cpus)) first = cpu;
                if (node_of_cpu(first) == node) home.push_back(&g);
            }
        }
        if (home.empty()) {
            const long pkg = package_of_cpu(sched_getcpu());
            for (const L3Group& g : l3_groups())
                if (g.package == pkg) home.push_back(&g);
        }
        if (trace) std::fprintf(stderr, "[svt] pack: records on NUMA node %ld, %zu L3 groups chosen\n", node, home.size());
    }
    // ---- how many: the encoder is a burst of a few ms per worker.  A cgroup CPU quota is CPU time per accounting period
    // (16 CPUs = 1.6 s per 100 ms), not a number of threads: a call whose whole work fits well inside one period's
    // allowance runs one worker per physical core next to the records (measured on 2 x EPYC 9575F, 1 M units: 16 / 32 / 64
    // / 128 workers -> 38 / 20 / 11 / 15 ms); a longer one is bound by the quota whatever it starts and keeps to
    // usable_cpus() (<= 16: the memory system of one socket does not feed more sustained workers any faster).
    unsigned want = std::min(usable_cpus(), 16u);
    {
        unsigned cores = 0;
        for (const L3Group* g : home) cores += g->cores;
        cores = std::min(cores, 64u);
        const CpuQuota q = cpu_quota();
        const double est_cpu_s = (double)n_rec_claimed * 6e-9 + (double)n * 50e-9;
        if (cores > want && (q.period_s == 0.0 || est_cpu_s <= 0.6 * q.cpu_s)) want = cores;
    }
    if (const char* e = std::getenv("SVT_PACK_THREADS")) want = (unsigned)std::max(1, std::atoi(e));   // (measurements)
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, n_chunks));
    std::vector<Worker> workers(nt);
    std::vector<ChunkOut> chunks(n_chunks);
    if (trace) std::fprintf(stderr, "[svt] pack: %u workers\n", nt);
    // ---- the one pass over the records: contract check + the three streams of every unit (record order).  Chunks are
    // claimed, not dealt: a worker that shares its core or loses its CPU for a while just takes fewer.
    std::vector<std::atomic<uint64_t>> next_chunk(n_ranges), next_copy(n_ranges);
    for (uint64_t r = 0; r < n_ranges; ++r) {
        next_chunk[r].store(r * chunks_per_range, std::memory_order_relaxed);
        next_copy[r].store(r * chunks_per_range, std::memory_order_relaxed);
    }
    std::function<void(Worker&, uint64_t)> encode_chunk;
    auto encode_phase = [&](uint64_t range, unsigned t) {
        const uint64_t range_end = std::min(n_chunks, (range + 1) * chunks_per_range);
        const auto w_t0 = std::chrono::steady_clock::now();
        // (worker 0 is the calling thread: its placement is the caller's business)
        if (range == 0 && spread && t > 0 && home.size() > 1) {
            const size_t n_groups = home.size();
            const cpu_set_t& g = home[t % n_groups]->cpus;
            cpu_set_t one = g;
            if (spread == 2) {
                // the (t / n_groups)-th CPU of the group: the low CPU numbers of a group are distinct cores, their SMT siblings follow
                CPU_ZERO(&one);
                unsigned want_k = t / (unsigned)n_groups, k = 0;
                for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu)
                    if (CPU_ISSET(cpu, &g) && k++ == want_k) { CPU_SET(cpu, &one); break; }
                if (CPU_COUNT(&one) == 0) one = g;
            }
            (void)pthread_setaffinity_np(pthread_self(), sizeof one, &one);
        }
        Worker& W = workers[t];
        // a guess at this worker's share (3.2 bytes per record is typical): growing later is only a copy
        if (range == 0) {
            W.arena = g_arenas.get();
            W.arena.reserve_first((size_t)(n_rec_claimed / nt / 4 * 3 / 2 + 4096));   // (chunks are claimed: shares differ; more comes in segments)
        }
        for (uint64_t ch; (ch = next_chunk[range].fetch_add(1, std::memory_order_relaxed)) < range_end;) encode_chunk(W, ch);
        W.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w_t0).count();
    };
    // one chunk of 256 units: contract check + the three streams of every unit (record order), appended to the worker's arena
    encode_chunk = [&](Worker& W, const uint64_t ch) {
        {
            ChunkOut& C = chunks[ch];
            W.arena.begin_chunk();
            const uint64_t u0 = ch * kChunkUnits, u1 = std::min(n, u0 + kChunkUnits);
            std::memcpy(out->units + u0, in->units + u0, (u1 - u0) * sizeof(svt_unit));
            for (uint64_t u = u0; u < u1; ++u) {
                const uint64_t r0 = in->rec_offset[u], r1 = in->rec_offset[u + 1];
                const svt_unit& U = in->units[u];
                int ue = kUnitOk;
                if (r1 < r0 || r1 > n_rec_claimed) ue = kUnitOffsets;
                else if (r1 - r0 > 0x3FFFFFFFull) ue = kUnitTooLong;
                else if (U.svtype > SVT_SVTYPE_BND) ue = kUnitSvtype;
                else if ((U.libs >> 24) != 0 || (U.flags & ~SVT_UNIT_SKIP)) ue = kUnitReserved;
                else if (U.var_length < -(1 << 30) || U.var_length > (1 << 30)) ue = kUnitVarLength;
                else if (U.svtype == SVT_SVTYPE_DEL && U.var_length < 0) ue = kUnitNegativeDel;
                if (ue != kUnitOk) {
                    if (!W.unit_error.load(std::memory_order_relaxed)) W.unit_error.store(ue, std::memory_order_relaxed);
                    off[3 * u + 1] = off[3 * u + 2] = off[3 * u + 3] = 0u;
                    continue;                      // (the batch is rejected; nothing of this unit is read)
                }
                // the unit's constants against a library's tables
                MultiUnit M;
                M.libs = libs;
                M.n_libs = n_libs;
                M.common = common;
                M.is_del = U.svtype == SVT_SVTYPE_DEL;
                M.vl = U.var_length;
                M.pos_delta = (double)U.pos_delta;
                if (multi && SVT_UNIT_LIBS_FIRST(U.libs) < n_libs) M.lo = SVT_UNIT_LIBS_FIRST(U.libs);     // (the hint, where there is one: the sample's first library)
                const uint64_t f = r1 - r0;
                // worst case per stream: every record a wide pair entry behind a pad half-word (3 half-words; several libraries:
                // and a library switch in front of it), one reference-read entry, two candidate entries; + two or three slots:
                // the vector form stores sixteen half-words at once
                const uint64_t cap_s = ((multi ? 4 : 3) * f + 8 + 7) / 8 + 3, cap_r = f / 7 + 4, cap_x = 2 * f / 7 + 2;
                if (W.scratch.size() < (cap_s + cap_r + cap_x) * 8) W.scratch.resize((cap_s + cap_r + cap_x) * 8);
                PairStream S(W.scratch.data());
                WeightStream R(W.scratch.data() + cap_s * 8), X(W.scratch.data() + (cap_s + cap_r) * 8);
                UnitState st;
                if (read_only_probe) {     // SVT_PACK_PROBE=1 (measurements): touch the unit's records and nothing else
                    uint32_t x = 0;
                    for (uint64_t j = r0; j < r1; ++j) x ^= recs[j].x ^ recs[j].w;
                    st.or_span = x & 0x7fffffffu;
                    st.or_flags = 0;
                } else if (!multi) {
                    const UnitCtx c = M.ctx_of(0);
#if SVT_PACK_AVX512
                    if (use_avx512) encode_records_avx512<false>(recs, r0, r1, c, nullptr, st, S, R, X);
                    else
#endif
                    encode_records(recs, r0, r1, c, st, S, R, X);
                } else {
                    // several libraries: a sample's reads come from one library as a rule, from two or three interleaved when it
                    // was sequenced more than once
#if SVT_PACK_AVX512
                    if (use_avx512) encode_records_avx512<true>(recs, r0, r1, M.ctx_of(0), &M, st, S, R, X);
                    else
#endif
                    encode_records_runs(recs, r0, r1, M, st, S, R, X);
                }
                const uint32_t lone = st.lone, or_flags = st.or_flags, or_span = st.or_span;
                const uint32_t bad_bits = (lone ? kErrStraddleNoPair : 0u) | ((multi ? M.bad_lib : (or_flags & 0xffff00u) != 0u) ? kErrLibIndex : 0u) |
                                          ((or_flags & ~SVT_REC_FLAG_MASK) ? kErrReservedBits : 0u) | ((int32_t)or_span < 0 ? kErrNegativeSpan : 0u);
                if (bad_bits) W.bad.fetch_or(bad_bits, std::memory_order_relaxed);
                const uint32_t ns = S.finish(), nr = R.finish(), nx = X.finish();
                off[3 * u + 1] = ns;
                off[3 * u + 2] = nr;
                off[3 * u + 3] = nx;
                Slot* dst = W.arena.append((size_t)ns + nr + nx);
                std::memcpy(dst, W.scratch.data(), (size_t)ns * 16);
                std::memcpy(dst + ns, W.scratch.data() + cap_s * 8, (size_t)nr * 16);
                std::memcpy(dst + ns + nr, W.scratch.data() + (cap_s + cap_r) * 8, (size_t)nx * 16);
            }
            C.src = W.arena.chunk_begin();
            C.n_slots = W.arena.chunk_size();
        }
    };
    // ---- between the phases, on the calling thread (svt_last_error is thread-local): verdict on the batch, slot counts
    // -> chunk bases, the output array
    uint64_t total = 0, range_first_slot = 0;
    Slot* slots = nullptr;
    auto between_phases = [&](uint64_t range) -> int {
        const uint64_t c0 = range * chunks_per_range, c1 = std::min(n_chunks, c0 + chunks_per_range);
        if (trace && range + 1 == n_ranges) {
            std::fprintf(stderr, "[svt] pack: worker ms:");
            for (const Worker& W : workers) std::fprintf(stderr, " %.1f", W.ms);
            std::fprintf(stderr, "\n");
        }
        if (range + 1 == n_ranges) mark("encode (one pass)");
        uint32_t bad = 0;
        int unit_error = kUnitOk;
        for (const Worker& W : workers) {
            bad |= W.bad.load(std::memory_order_relaxed);
            if (W.unit_error.load(std::memory_order_relaxed) && !unit_error) unit_error = W.unit_error.load(std::memory_order_relaxed);
        }
        switch (unit_error) {
        case kUnitOffsets: return fail(SVT_ERR_INVALID, "rec_offset not monotone");
        case kUnitTooLong: return fail(SVT_ERR_INVALID, "unit with too many records");
        case kUnitSvtype: return fail(SVT_ERR_INVALID, "bad svtype");
        case kUnitReserved: return fail(SVT_ERR_INVALID, "unit reserved/flags bits must be 0");
        case kUnitVarLength: return fail(SVT_ERR_UNSUPPORTED, "var_length outside the packed format's range");
        case kUnitNegativeDel: return fail(SVT_ERR_UNSUPPORTED, "negative DEL length");
        default: break;
        }
        if (bad) return fail(SVT_ERR_INVALID, record_error_text(bad));

        // slot counts -> slot offsets: chunk bases serially, inside a chunk in parallel (second phase)
        range_first_slot = total;
        for (uint64_t ch = c0; ch < c1; ++ch) {
            ChunkOut& C = chunks[ch];
            C.base = total;
            total += C.n_slots;
            if (total >= 0xFFFFFFF0ull) return fail(SVT_ERR_UNSUPPORTED, "too many slots for 32-bit slot offsets");
        }
        if (sink) {
            if (total > sink->slots_cap) return SVT_ERR_PACK_OVERFLOW;   // (the caller repeats the call without a sink)
        } else {
            out->slots = A.get(std::max<uint64_t>(total, 1) * 16);
            if (!out->slots) return fail(SVT_ERR_NOMEM, "out of host memory");
            mark("allocate slots");
        }
        slots = static_cast<Slot*>(out->slots);
        return SVT_OK;
    };
    auto copy_phase = [&](uint64_t range, unsigned) {
        const uint64_t range_end = std::min(n_chunks, (range + 1) * chunks_per_range);
        for (uint64_t ch; (ch = next_copy[range].fetch_add(1, std::memory_order_relaxed)) < range_end;) {
            const ChunkOut& C = chunks[ch];
            const uint64_t u0 = ch * kChunkUnits, u1 = std::min(n, u0 + kChunkUnits);
            uint64_t run = C.base;
            for (uint64_t i = 3 * u0 + 1; i <= 3 * u1; ++i) {
                run += off[i];
                off[i] = (uint32_t)run;
            }
            if (!C.n_slots) continue;
#if SVT_PACK_AVX512
            if (use_avx512) copy_streaming(slots + C.base, C.src, (size_t)C.n_slots * 16);
            else
#endif
            std::memcpy(slots + C.base, C.src, (size_t)C.n_slots * 16);
        }
    };
    auto hand_over = [&](uint64_t range) -> int {
        if (!sink || !sink->ready) return SVT_OK;
        const uint64_t u0 = std::min(n, range * chunks_per_range * kChunkUnits), u1 = std::min(n, (range + 1) * chunks_per_range * kChunkUnits);
        return sink->ready(sink->ctx, out, u0, u1, range_first_slot, total);
    };
    if (sink && sink->ready && n_chunks > 0 && !std::getenv("SVT_PACK_MEETINGS")) {
        // ---- the streamed form: no meetings.  Workers claim chunks from ONE counter, in order; whoever encodes the last chunk of a
        // range becomes its finisher -- waits for the range before it to have its slot bases, fixes this range's, and copies the
        // range into the final arrays (helped by whoever has nothing left to encode); the calling thread does not encode at
        // all: it hands finished ranges over, in order, while the workers are far ahead.  (With meetings every range cost the
        // threads three rendezvous, each as slow as the slowest -- possibly throttled -- thread.)
        struct Range {
            std::atomic<uint32_t> encoded{0}, next_copy{0}, copied{0};
            std::atomic<int> state{0};          // 0 pending, 1 bases fixed (copy open), 2 copied, -1 failed
            uint64_t first_slot = 0, end_slot = 0;
        };
        std::vector<Range> ranges(n_ranges);
        auto chunks_in = [&](uint64_t r) { return (uint32_t)(std::min(n_chunks, (r + 1) * chunks_per_range) - r * chunks_per_range); };
        std::atomic<uint64_t> claim{0};
        std::atomic<int> failed{0};             // 0 none, else the code below
        enum { kFailVerdict = 1, kFailOverflow, kFailTooMany, kFailException };
        std::exception_ptr thrown;
        std::mutex lock;
        slots = static_cast<Slot*>(out->slots);
        auto nap = [](unsigned& spins) {
            if (++spins < 2048) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            } else std::this_thread::sleep_for(std::chrono::microseconds(20));
        };
        auto copy_some = [&](uint64_t r) -> bool {     // one chunk of range r into the final arrays; false: nothing left to claim
            Range& G = ranges[r];
            const uint32_t k = G.next_copy.fetch_add(1, std::memory_order_relaxed), total_k = chunks_in(r);
            if (k >= total_k) return false;
            const uint64_t ch = r * chunks_per_range + k;
            const ChunkOut& C = chunks[ch];
            const uint64_t u0 = ch * kChunkUnits, u1 = std::min(n, u0 + kChunkUnits);
            uint64_t run = C.base;
            for (uint64_t i = 3 * u0 + 1; i <= 3 * u1; ++i) {
                run += off[i];
                off[i] = (uint32_t)run;
            }
            if (C.n_slots) {
#if SVT_PACK_AVX512
                if (use_avx512) copy_streaming(slots + C.base, C.src, (size_t)C.n_slots * 16);
                else
#endif
                std::memcpy(slots + C.base, C.src, (size_t)C.n_slots * 16);
            }
            if (G.copied.fetch_add(1, std::memory_order_acq_rel) + 1 == total_k) G.state.store(2, std::memory_order_release);
            return true;
        };
        auto finish = [&](uint64_t r) {                // the range's last chunk has been encoded (by this thread)
            unsigned spins = 0;
            while (r > 0 && ranges[r - 1].state.load(std::memory_order_acquire) == 0 && !failed.load(std::memory_order_relaxed)) nap(spins);
            if (failed.load(std::memory_order_relaxed) || (r > 0 && ranges[r - 1].state.load(std::memory_order_acquire) < 0)) {
                ranges[r].state.store(-1, std::memory_order_release);
                return;
            }
            int why = 0;
            for (const Worker& W : workers)
                if (W.bad.load(std::memory_order_relaxed) || W.unit_error.load(std::memory_order_relaxed)) why = kFailVerdict;
            uint64_t at = r ? ranges[r - 1].end_slot : 0;
            ranges[r].first_slot = at;
            const uint64_t c0 = r * chunks_per_range, c1 = c0 + chunks_in(r);
            for (uint64_t ch = c0; ch < c1 && !why; ++ch) {
                chunks[ch].base = at;
                at += chunks[ch].n_slots;
                if (at >= 0xFFFFFFF0ull) why = kFailTooMany;
            }
            if (!why && at > sink->slots_cap) why = kFailOverflow;
            if (why) {
                int none = 0;
                failed.compare_exchange_strong(none, why);
                ranges[r].state.store(-1, std::memory_order_release);
                return;
            }
            ranges[r].end_slot = at;
            ranges[r].state.store(1, std::memory_order_release);
            while (copy_some(r)) {}
        };
        auto worker = [&](unsigned t) {
            try {
                const auto w_t0 = std::chrono::steady_clock::now();
                Worker& W = workers[t];
                W.arena = g_arenas.get();
                W.arena.reserve_first((size_t)(n_rec_claimed / nt / 4 * 3 / 2 + 4096));
                for (uint64_t ch; !failed.load(std::memory_order_relaxed) && (ch = claim.fetch_add(1, std::memory_order_relaxed)) < n_chunks;) {
                    encode_chunk(W, ch);
                    const uint64_t r = ch / chunks_per_range;
                    if (ranges[r].encoded.fetch_add(1, std::memory_order_acq_rel) + 1 == chunks_in(r)) finish(r);
                }
                // nothing left to encode: help with the copies of the ranges that are open, oldest first, until all are through
                for (uint64_t r = 0; r < n_ranges && !failed.load(std::memory_order_relaxed);) {
                    const int st = ranges[r].state.load(std::memory_order_acquire);
                    if (st < 0) break;
                    if (st == 2) { ++r; continue; }
                    if (st == 1 && copy_some(r)) continue;
                    unsigned sp = 2048;          // bases not fixed yet, or every copy of the range is claimed and some are still running
                    nap(sp);
                }
                W.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w_t0).count();
            } catch (...) {
                {
                    std::lock_guard<std::mutex> g(lock);
                    if (!thrown) thrown = std::current_exception();
                }
                int none = 0;
                failed.compare_exchange_strong(none, kFailException);
            }
        };
        std::vector<std::thread> pool;
        pool.reserve(nt);
        for (unsigned t = 0; t < nt; ++t) {
            try {
                pool.emplace_back(worker, t);
            } catch (const std::system_error&) {
                break;
            }
        }
        int rc = SVT_OK;
        if (pool.empty()) worker(0);             // (no thread could be started: the caller encodes, then hands over)
        for (uint64_t r = 0; r < n_ranges && rc == SVT_OK; ++r) {
            unsigned spins = 0;
            int st;
            while ((st = ranges[r].state.load(std::memory_order_acquire)) != 2 && st >= 0 && !failed.load(std::memory_order_relaxed)) nap(spins);
            if (ranges[r].state.load(std::memory_order_acquire) != 2) break;
            const uint64_t u0 = std::min(n, r * chunks_per_range * kChunkUnits), u1 = std::min(n, (r + 1) * chunks_per_range * kChunkUnits);
            rc = sink->ready(sink->ctx, out, u0, u1, ranges[r].first_slot, ranges[r].end_slot);
            if (rc != SVT_OK) {
                int none = 0;
                failed.compare_exchange_strong(none, kFailException + 1);   // (the consumer declined: stop the workers)
            }
        }
        for (auto& th : pool) th.join();
        if (thrown) std::rethrow_exception(thrown);
        if (rc != SVT_OK) return rc;
        switch (failed.load()) {
        case 0: break;
        case kFailOverflow: return SVT_ERR_PACK_OVERFLOW;
        case kFailTooMany: return fail(SVT_ERR_UNSUPPORTED, "too many slots for 32-bit slot offsets");
        default: {
            const int v = between_phases(n_ranges - 1);          // the verdict's text, on the calling thread (svt_last_error is thread-local)
            return v != SVT_OK ? v : fail(SVT_ERR_INTERNAL, "the encoder stopped without a verdict");
        }
        }
        total = ranges[n_ranges - 1].end_slot;
        mark("encode + copy (streamed)");
        out->n_slots = total;
        out->n_records = n_rec_claimed;
        release.armed = false;
        return SVT_OK;
    }
    SVT_TRY(run_ranged_phases(nt, n_ranges, encode_phase, between_phases, copy_phase, hand_over));
    mark("offsets + final copy");
    out->n_slots = total;
    out->n_records = n_rec_claimed;
    release.armed = false;
    return SVT_OK;
}

}  // namespace svt
