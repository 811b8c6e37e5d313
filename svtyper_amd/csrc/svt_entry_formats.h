// svt_entry_formats.h -- the entries of packed evidence (include/svtyper_hip.h: svt_packed_evidence) and their encoders
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
//
// svt_pack_evidence (host, svtyper_hip.hip) writes them, svt_packed_kernel.h / svt_unit_math.h consume them,
// oracle/py_packed.py decodes them independently on the CPU.
#ifndef SVT_ENTRY_FORMATS_H
#define SVT_ENTRY_FORMATS_H

#include "svt_device_types.h"

namespace svt {

// A unit's evidence becomes three sparse streams of entries packed into 16-byte slots, each in record order.
// The five tallies are independent sums, so evidence for different tallies can live in different streams
// without changing any of them.
//
//   pair entries (alt_span, ref_span), one library with n_bins <= 2047, in half-words:
//     an entry with the batch's most common MAPQ pair (60, 60 for bwa) is ONE half-word f3 | code << 3; any other
//     entry is a 4-byte-aligned pair of half-words, f3 | code << 3 | 0x8000 then mapq_a | mapq_b << 8.  A zero
//     half-word is a no-op (f3 = 0: both weights 0) and pads a wide entry to its alignment.
//     f3   = alt | refA << 1 | refB << 2 straddle bits
//     code = ospan_len translated into the index space of the library's histogram tables: with
//            r = ospan_len - key_min, the kernel needs thr[r] (parsers.py:870-872) and, for a
//            DEL, hist[r - var_length] (parsers.py:874-878), each replaced by the sentinel bin
//            n_bins when out of range.  With off2 = min(var_length, n_bins):
//                var_length <  n_bins:  code = r            for 0 <= r < var_length + n_bins
//                var_length >= n_bins:  code = r            for 0 <= r < n_bins
//                                       code = n_bins + (r - var_length)   for 0 <= r - var_length < n_bins
//                anything else / not a DEL window:  code = 2 * n_bins
//            so that i1 = min(code, n_bins) and i2 = min(code - off2, n_bins) (unsigned) are exactly the
//            two table indices.  Only the addressing is precomputed; the look-ups, the p_concordant
//            decision and every sum stay in the kernel.
//   reference-read entries (ref_seq)         2 bytes: mapq0, mapq1 -- seven per slot (bytes 0..13); byte 14
//                                            of the slot holds their seven first_of_fragment bits
//   candidate entries (alt_seq / alt_clip)   the same, plus their seven is_clip bits in byte 15
//     the two gated MAPQs of the reference reads (rs_a, rs_b), of the split candidate (seq_l, seq_r)
//     or of the clip candidate (clip_l, clip_r); first_of_fragment marks the first kept entry for its
//     tally in a read-fragment (sso association: fragment-local sums).
//
// Entries that can only add +0.0 to a sum are not stored: pair entries without a straddle bit, with a
// zero MAPQ on either read, or of a DEL smaller than 2 sd of the library (classic.py:339,383);
// weight entries whose two gated MAPQs are 0.  x + 0.0 == x bit-for-bit for these non-negative sums.
// Order inside a stream is record order, so every sum sees the reference's additions in the reference's order.

struct UnitGeom {
    bool is_del;
    int32_t var_length;
    double pos_delta_d;
};

inline UnitGeom unit_geom(const svt_unit& U)
{
    UnitGeom g;
    g.is_del = U.svtype == SVT_SVTYPE_DEL;
    g.var_length = U.var_length;
    g.pos_delta_d = (double)U.pos_delta;
    return g;
}

inline bool keeps_pair_entry(const uint4 w, const UnitGeom& g, const LibDesc& lib)
{
    if ((w.w & 7u) == 0u) return false;
    if ((w.y & 0xffu) == 0u || (w.y & 0xff00u) == 0u) return false;      // prob_mapq(0) == 0.0
    if (g.is_del && g.pos_delta_d < lib.sd2) return false;                // classic.py:339,383
    return true;
}

inline uint32_t pair_code(const uint32_t ospan_len, const UnitGeom& g, const LibDesc& lib)
{
    const int64_t nb = lib.n_bins;
    const int64_t r = (int64_t)(int32_t)ospan_len - (int64_t)lib.key_min;
    const bool in1 = r >= 0 && r < nb;
    if (!g.is_del) return in1 ? (uint32_t)r : (uint32_t)(2 * nb);
    const int64_t vl = g.var_length;            // >= 0 (checked on the host before this layout is chosen)
    const int64_t r2 = r - vl;
    const bool in2 = r2 >= 0 && r2 < nb;
    if (vl < nb) return (r >= 0 && r < vl + nb) ? (uint32_t)r : (uint32_t)(2 * nb);
    return in1 ? (uint32_t)r : in2 ? (uint32_t)(nb + r2) : (uint32_t)(2 * nb);
}

// the gated MAPQ pairs (lo byte, hi byte) of a canonical record that feed ref_seq / alt_seq / alt_clip;
// 0 = nothing to add
inline void weight_pairs(const uint4 w, uint32_t k[3])
{
    k[0] = w.y >> 16;            // rs_a | rs_b << 8
    k[1] = w.z & 0xffffu;        // seq_l | seq_r << 8
    k[2] = w.z >> 16;            // clip_l | clip_r << 8
}

constexpr uint32_t kWideEntry = 0x8000u;      // pair stream: the next half-word holds this entry's MAPQs
constexpr uint32_t kDefaultCommonMapq = 60u | (60u << 8);
constexpr uint32_t kVoteRecords = 1u << 14;   // records the (host-side) MAPQ vote looks at

// seven 2-byte MAPQ-pair entries per 16-byte slot: bytes 0..13; bit k of byte 14 = entry k
// is the first kept one of its fragment; bit k of byte 15 = entry k is a clip candidate (candidate stream)
struct WeightRowWriter {
    uint4* out;       // row 0 of this lane
    uint32_t n = 0;   // entries so far
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    uint32_t stride = 1;   // slots between consecutive slots of the stream
    inline void put(const uint32_t mapq_pair, const bool first, const bool clip = false)
    {
        const uint32_t k = n % 7u;
        const uint32_t half = mapq_pair << ((k & 1u) * 16u);
        switch (k >> 1) {
        case 0: w[0] |= half; break;
        case 1: w[1] |= half; break;
        case 2: w[2] |= half; break;
        default: w[3] |= half;        // k == 6: low half of the last dword
        }
        if (first) w[3] |= 1u << (16u + k);
        if (clip) w[3] |= 1u << (24u + k);
        if (k == 6u) {
            out[(uint64_t)(n / 7u) * stride] = make_uint4(w[0], w[1], w[2], w[3]);
            w[0] = w[1] = w[2] = w[3] = 0u;
        }
        ++n;
    }
    inline void finish(const uint32_t rows)
    {
        uint32_t r = n / 7u;
        if (n % 7u) out[(uint64_t)r++ * stride] = make_uint4(w[0], w[1], w[2], w[3]);
        for (; r < rows; ++r) out[(uint64_t)r * stride] = make_uint4(0, 0, 0, 0);
    }
};

// pair stream: eight half-words per 16-byte slot
struct ShortRowWriter {
    uint4* out;       // row 0 of this lane
    uint32_t n = 0;   // half-words so far
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    uint32_t stride = 1;
    inline void put_half(const uint32_t hw)
    {
        const uint32_t k = n & 7u;
        const uint32_t v = hw << ((k & 1u) * 16u);
        switch (k >> 1) {
        case 0: w[0] |= v; break;
        case 1: w[1] |= v; break;
        case 2: w[2] |= v; break;
        default: w[3] |= v;
        }
        if (k == 7u) {
            out[(uint64_t)(n >> 3) * stride] = make_uint4(w[0], w[1], w[2], w[3]);
            w[0] = w[1] = w[2] = w[3] = 0u;
        }
        ++n;
    }
    inline void put(const uint32_t lo16, const uint32_t mq, const uint32_t common)
    {
        if (mq == common) {
            put_half(lo16);
        } else {
            if (n & 1u) put_half(0u);          // no-op: wide entries start on a 4-byte boundary
            put_half(lo16 | kWideEntry);
            put_half(mq);
        }
    }
    inline void finish(const uint32_t rows)
    {
        uint32_t r = n >> 3;
        if (n & 7u) out[(uint64_t)r++ * stride] = make_uint4(w[0], w[1], w[2], w[3]);
        for (; r < rows; ++r) out[(uint64_t)r * stride] = make_uint4(0, 0, 0, 0);
    }
};

}  // namespace svt

#endif  // SVT_ENTRY_FORMATS_H
