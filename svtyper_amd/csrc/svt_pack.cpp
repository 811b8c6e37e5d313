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
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

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
    explicit PairStream(uint16_t* p) : begin(p) {}
    inline void put(const uint32_t lo16, const uint32_t mq, const uint32_t common)
    {
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

#ifndef SVT_PACK_BRANCHLESS_REF
#define SVT_PACK_BRANCHLESS_REF 1
#endif
constexpr uint64_t kChunkUnits = 256;
static_assert(SVT_REC_CONTINUATION == (1u << 3) && SVT_REC_HAS_PAIR == (1u << 4), "bit positions used by the encoder's loop");

struct ChunkOut {          // where a chunk's slots wait for the final copy
    unsigned worker = 0;
    uint64_t arena_at = 0; // first slot in the worker's arena
    uint64_t n_slots = 0;
    uint64_t base = 0;     // first slot in the final array
};

struct Worker {
    std::vector<Slot> arena;       // the slots of this worker's chunks, chunk after chunk
    std::vector<uint16_t> scratch; // one unit's three streams at worst-case size, as half-words
    uint32_t bad = 0;              // record-contract bits (kErr*)
    int unit_error = 0;            // first unit-array violation (1-based code below), 0 = none
};

enum UnitError { kUnitOk = 0, kUnitOffsets, kUnitTooLong, kUnitSvtype, kUnitReserved, kUnitVarLength, kUnitNegativeDel };

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

int encode_packed(const svt_evidence_batch* in, const PackAlloc& A, PackedArrays* out)
{
    if (!in || !out) return fail(SVT_ERR_INVALID, "null argument");
    *out = PackedArrays{};
    const uint64_t n = in->n_units;
    if (n >= 0x55555550ull) return fail(SVT_ERR_INVALID, "too many units in one batch");
    if (in->n_libs == 0 || in->n_libs > 256 || !in->libs) return fail(SVT_ERR_INVALID, "n_libs must be 1..256");
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
    if (in->n_libs != 1) return fail(SVT_ERR_UNSUPPORTED, "packed evidence holds one library");
    if (T.libs[0].n_bins > kMaxShortBins) return fail(SVT_ERR_UNSUPPORTED, "histogram too wide for the packed pair entries");
    if (!T.fast_geometry) return fail(SVT_ERR_UNSUPPORTED, "library geometry outside the packed format's range");
    const LibDesc lib = T.libs[0];
    const int64_t key_min = lib.key_min, nb = lib.n_bins;
    const Slot* recs = reinterpret_cast<const Slot*>(in->records);
    const uint64_t n_rec_claimed = n ? in->rec_offset[n] : 0;
    if (n_rec_claimed && !in->records) return fail(SVT_ERR_INVALID, "null records");

    struct Release {
        const PackAlloc& A;
        PackedArrays* p;
        bool armed = true;
        ~Release() { if (armed) { A.put(p->off); A.put(p->units); A.put(p->slots); *p = PackedArrays{}; } }
    } release{A, out};
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
    unsigned want = std::min(usable_cpus(), 16u);
    if (const char* e = std::getenv("SVT_PACK_THREADS")) want = (unsigned)std::max(1, std::atoi(e));   // (measurements)
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, n_chunks));
    std::vector<Worker> workers(nt);
    std::vector<ChunkOut> chunks(n_chunks);

    // ---- the one pass over the records: contract check + the three streams of every unit (record order)
    run_threads(nt, [&](unsigned t) {
        Worker& W = workers[t];
        // a guess at this worker's share (3.2 bytes per record is typical): growing later is only a copy
        W.arena.reserve((size_t)(n_rec_claimed / nt / 4 + 4096));
        for (uint64_t ch = t; ch < n_chunks; ch += nt) {
            ChunkOut& C = chunks[ch];
            C.worker = t;
            C.arena_at = W.arena.size();
            const uint64_t u0 = ch * kChunkUnits, u1 = std::min(n, u0 + kChunkUnits);
            std::memcpy(out->units + u0, in->units + u0, (u1 - u0) * sizeof(svt_unit));
            for (uint64_t u = u0; u < u1; ++u) {
                const uint64_t r0 = in->rec_offset[u], r1 = in->rec_offset[u + 1];
                const svt_unit& U = in->units[u];
                int ue = kUnitOk;
                if (r1 < r0 || r1 > n_rec_claimed) ue = kUnitOffsets;
                else if (r1 - r0 > 0x3FFFFFFFull) ue = kUnitTooLong;
                else if (U.svtype > SVT_SVTYPE_BND) ue = kUnitSvtype;
                else if ((U.libs >> 16) != 0 || (U.flags & ~SVT_UNIT_SKIP)) ue = kUnitReserved;
                else if (U.var_length < -(1 << 30) || U.var_length > (1 << 30)) ue = kUnitVarLength;
                else if (U.svtype == SVT_SVTYPE_DEL && U.var_length < 0) ue = kUnitNegativeDel;
                if (ue != kUnitOk) {
                    if (!W.unit_error) W.unit_error = ue;
                    off[3 * u + 1] = off[3 * u + 2] = off[3 * u + 3] = 0u;
                    continue;                      // (the batch is rejected; nothing of this unit is read)
                }
                const bool is_del = U.svtype == SVT_SVTYPE_DEL;
                const bool gated = is_del && (double)U.pos_delta < lib.sd2;            // classic.py:339,383: no pair entry adds anything
                // pair_code (svt_entry_formats.h) with the unit's constants folded: code = r when (uint64) r < lim1, else
                // nb + (r - vl) when (uint64)(r - vl) < lim2, else 2 nb          (r = ospan_len - key_min)
                const int64_t vl = U.var_length;
                const uint64_t lim1 = !is_del ? (uint64_t)nb : vl < nb ? (uint64_t)(vl + nb) : (uint64_t)nb;
                const uint64_t lim2 = is_del && vl >= nb ? (uint64_t)nb : 0u;
                const uint32_t code_out = (uint32_t)(2 * nb);
                const uint64_t f = r1 - r0;
                // worst case per stream: every record a wide pair entry behind a pad half-word (3 half-words), one
                // reference-read entry, two candidate entries
                const uint64_t cap_s = (3 * f + 8 + 7) / 8 + 1, cap_r = f / 7 + 2, cap_x = 2 * f / 7 + 2;   // (+ room for the unconditional writes)
                if (W.scratch.size() < (cap_s + cap_r + cap_x) * 8) W.scratch.resize((cap_s + cap_r + cap_x) * 8);
                PairStream S(W.scratch.data());
                WeightStream R(W.scratch.data() + cap_s * 8), X(W.scratch.data() + (cap_s + cap_r) * 8);
                bool has_r = false, has_s = false, has_c = false;    // the fragment already has a kept entry for that tally
                uint32_t or_flags = 0, or_span = 0, lone = 0;        // the record contract, folded like the device's RecordCheck
                // (measured: writing every entry unconditionally and advancing by 0 / 1 instead of branching is slower --
                // the loop then retires seven stores per record; the branches below are mostly predictable)
                for (uint64_t j = r0; j < r1; ++j) {
                    const Slot w = recs[j];
                    const uint32_t fl = w.w;
                    or_flags |= fl;
                    or_span |= w.x;
                    lone |= (fl & 7u) & (((fl >> 4) & 1u) - 1u);     // straddle bits of a record without HAS_PAIR
                    if (!(fl & SVT_REC_CONTINUATION)) has_r = has_s = has_c = false;
                    // a pair entry that could only add +0.0 is not stored: no straddle bit, a zero MAPQ (prob_mapq(0) == 0.0), a gated DEL
                    if ((fl & 7u) && (w.y & 0xffu) && (w.y & 0xff00u) && !gated) {
                        const int64_t r = (int64_t)(int32_t)w.x - key_min;
                        const uint32_t code = (uint64_t)r < lim1 ? (uint32_t)r : (uint64_t)(r - vl) < lim2 ? (uint32_t)(nb + r - vl) : code_out;
                        S.put((fl & 7u) | (code << 3), w.y & 0xffffu, common);
                    }
                    const uint32_t k_ref = w.y >> 16, k_seq = w.z & 0xffffu, k_clip = w.z >> 16;   // gated MAPQ pairs; 0 = nothing to add
#if SVT_PACK_BRANCHLESS_REF
                    R.put_if(k_ref, k_ref ? 1u : 0u, has_r ? 0u : 1u);
                    has_r |= k_ref != 0u;
#else
                    if (k_ref) { R.put(k_ref, !has_r, false); has_r = true; }
#endif
                    if (k_seq) { X.put(k_seq, !has_s, false); has_s = true; }
                    if (k_clip) { X.put(k_clip, !has_c, true); has_c = true; }
                }
                W.bad |= (lone ? kErrStraddleNoPair : 0u) | ((or_flags & 0xff00u) ? kErrLibIndex : 0u) |
                         ((or_flags & ~SVT_REC_FLAG_MASK) ? kErrReservedBits : 0u) | ((int32_t)or_span < 0 ? kErrNegativeSpan : 0u);
                const uint32_t ns = S.finish(), nr = R.finish(), nx = X.finish();
                off[3 * u + 1] = ns;
                off[3 * u + 2] = nr;
                off[3 * u + 3] = nx;
                const size_t at = W.arena.size();
                if (W.arena.capacity() < at + ns + nr + nx) W.arena.reserve(2 * W.arena.capacity() + ns + nr + nx);
                W.arena.resize(at + ns + nr + nx);
                std::memcpy(W.arena.data() + at, W.scratch.data(), (size_t)ns * 16);
                std::memcpy(W.arena.data() + at + ns, W.scratch.data() + cap_s * 8, (size_t)nr * 16);
                std::memcpy(W.arena.data() + at + ns + nr, W.scratch.data() + (cap_s + cap_r) * 8, (size_t)nx * 16);
            }
            C.n_slots = W.arena.size() - C.arena_at;
        }
    });
    mark("encode (one pass)");
    uint32_t bad = 0;
    int unit_error = kUnitOk;
    for (const Worker& W : workers) {
        bad |= W.bad;
        if (W.unit_error && !unit_error) unit_error = W.unit_error;
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

    // ---- slot counts -> slot offsets: chunk bases serially, inside a chunk in parallel
    uint64_t total = 0;
    for (ChunkOut& C : chunks) {
        C.base = total;
        total += C.n_slots;
        if (total >= 0xFFFFFFF0ull) return fail(SVT_ERR_UNSUPPORTED, "too many slots for 32-bit slot offsets");
    }
    out->slots = A.get(std::max<uint64_t>(total, 1) * 16);
    if (!out->slots) return fail(SVT_ERR_NOMEM, "out of host memory");
    Slot* slots = static_cast<Slot*>(out->slots);
    mark("allocate slots");
    run_threads(nt, [&](unsigned t) {
        for (uint64_t ch = t; ch < n_chunks; ch += nt) {
            const ChunkOut& C = chunks[ch];
            const uint64_t u0 = ch * kChunkUnits, u1 = std::min(n, u0 + kChunkUnits);
            uint64_t run = C.base;
            for (uint64_t i = 3 * u0 + 1; i <= 3 * u1; ++i) {
                run += off[i];
                off[i] = (uint32_t)run;
            }
            std::memcpy(slots + C.base, workers[C.worker].arena.data() + C.arena_at, (size_t)C.n_slots * 16);
        }
    });
    mark("offsets + final copy");
    out->n_slots = total;
    out->n_records = n_rec_claimed;
    release.armed = false;
    return SVT_OK;
}

}  // namespace svt
