// svt_prepare_kernels.h -- scan / re-tile kernels that run once per batch
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_PREPARE_KERNELS_H
#define SVT_PREPARE_KERNELS_H

#include "svt_device_types.h"

namespace svt {

// ------------------------------------------------------------------------------------------
// batch preparation kernels (run once per batch, outside the genotyping pass)
// ------------------------------------------------------------------------------------------

// ---- compact layout --------------------------------------------------------------------------
// A unit's evidence becomes three sparse streams of entries packed into 16-byte row slots, each in
// record order.  The five tallies are independent sums, so evidence for different tallies can live
// in different streams without changing any of them.
//
//   pair entry (alt_span, ref_span)
//       one library:  f3 | code << 3 | mapq_a << 16 | mapq_b << 24
//       several:      f3 | code << 3 | (lib - lib_min) << 16 | mapq_a << 18 | mapq_b << 25
//     f3   = alt | refA << 1 | refB << 2 straddle bits
//     code = ospan_len translated into the index space of the library's histogram tables: with
//            r = ospan_len - key_min, the kernel needs bins[r].thr (parsers.py:870-872) and, for a
//            DEL, bins[r - var_length].hist (parsers.py:874-878), each replaced by the sentinel bin
//            n_bins when out of range.  With off2 = min(var_length, n_bins):
//                var_length <  n_bins:  code = r            for 0 <= r < var_length + n_bins
//                var_length >= n_bins:  code = r            for 0 <= r < n_bins
//                                       code = n_bins + (r - var_length)   for 0 <= r - var_length < n_bins
//                anything else / not a DEL window:  code = 2 * n_bins
//            so that i1 = min(code, n_bins) and i2 = min(code - off2, n_bins) (unsigned) are exactly the
//            two table indices.  Only the addressing is precomputed; the look-ups, the p_concordant
//            decision and every sum stay in the genotype kernel.  `code << 3` is the byte offset of
//            the bin, and the MAPQs sit on byte boundaries, so the kernel turns every field into an LDS
//            address with one instruction.
//   reference-read entries (ref_seq)         2 bytes: mapq0, mapq1 -- seven per row slot (bytes 0..13); byte 14
//                                            of the slot holds their seven first_of_fragment bits
//   candidate entries (alt_seq / alt_clip)   the same, plus their seven is_clip bits in byte 15
//     the two gated MAPQs of the reference reads (rs_a, rs_b), of the split candidate (seq_l, seq_r)
//     or of the clip candidate (clip_l, clip_r); first_of_fragment marks the first kept entry for its
//     tally in a read-fragment (sso association: fragment-local sums).
//
//   short layout (one library, n_bins <= 2047, so `code` needs 12 bits and bit 15 of the low half is free):
//     most pair entries of a real batch carry the same two MAPQs (60, 60 for bwa), so the pair stream is
//     written in half-words.  An entry with the batch's most common MAPQ pair is ONE half-word
//     f3 | code << 3; any other entry is a 4-byte-aligned pair of half-words, f3 | code << 3 | 0x8000 then
//     mapq_a | mapq_b << 8.  A zero half-word is a no-op (f3 = 0: both weights 0) and pads a wide entry to
//     its alignment.  Order is untouched, so every sum sees the same additions in the same order.
//
// Entries that can only add +0.0 to a sum are not stored: pair entries without a straddle bit, with a
// zero MAPQ on either read, or of a DEL smaller than 2 sd of the entry's library (classic.py:339,383);
// weight entries whose two gated MAPQs are 0.  x + 0.0 == x bit-for-bit for these non-negative sums.

struct UnitGeom {
    bool is_del;
    int32_t var_length;
    double pos_delta_d;
};

__host__ __device__ __forceinline__ UnitGeom unit_geom(const svt_unit& U)
{
    UnitGeom g;
    g.is_del = U.svtype == SVT_SVTYPE_DEL;
    g.var_length = U.var_length;
    g.pos_delta_d = (double)U.pos_delta;
    return g;
}

__host__ __device__ __forceinline__ bool keeps_pair_entry(const uint4 w, const UnitGeom& g, const LibDesc& lib)
{
    if ((w.w & 7u) == 0u) return false;
    if ((w.y & 0xffu) == 0u || (w.y & 0xff00u) == 0u) return false;      // prob_mapq(0) == 0.0
    if (g.is_del && g.pos_delta_d < lib.sd2) return false;                // classic.py:339,383
    return true;
}

__host__ __device__ __forceinline__ uint32_t pair_code(const uint32_t ospan_len, const UnitGeom& g, const LibDesc& lib)
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
__host__ __device__ __forceinline__ void weight_pairs(const uint4 w, uint32_t k[3])
{
    k[0] = w.y >> 16;            // rs_a | rs_b << 8
    k[1] = w.z & 0xffffu;        // seq_l | seq_r << 8
    k[2] = w.z >> 16;            // clip_l | clip_r << 8
}

struct ScanOut {          // per unit
    uint32_t n[kStreams]; // entries per compact stream
    uint32_t libs;        // lib_min | lib_max << 8
    uint32_t flags;       // kScan*
    uint32_t n_short;     // half-words of the pair stream in the short layout
};
constexpr uint32_t kScanWideMapq = 1u;   // a kept pair entry has a MAPQ > 127
constexpr uint32_t kWideEntry = 0x8000u; // short layout: the next half-word holds this entry's MAPQs
constexpr uint32_t kDefaultCommonMapq = 60u | (60u << 8);
constexpr uint32_t kVoteRecords = 1u << 14;   // records the (host-side) MAPQ vote looks at

struct ScanArgs {
    const uint4* csr;
    const uint64_t* rec_offset;
    const svt_unit* units;
    const LibDesc* libs;
    uint64_t n_units;
    uint32_t n_libs;
    ScanOut* out;
    uint32_t* err;
    const uint32_t* common_mq;   // device word: the vote's result
};

// one thread per unit: validate the record contract of include/svtyper_hip.h, count the entries of the
// compact streams and find the range of libraries the unit references
__global__ __launch_bounds__(kBlock) void svt_scan_kernel(const ScanArgs a)
{
    const uint64_t u = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (u >= a.n_units) return;
    const uint64_t lo = a.rec_offset[u], hi = a.rec_offset[u + 1];
    const UnitGeom g = unit_geom(a.units[u]);
    ScanOut o{};
    const uint32_t common = *a.common_mq;
    uint32_t bad = 0, lib_min = 0xffu, lib_max = 0u;
    for (uint64_t j = lo; j < hi; ++j) {
        const uint4 w = a.csr[j];
        const uint32_t f = w.w;
        const uint32_t lib = SVT_REC_LIB(f);
        lib_min = min(lib_min, lib);
        lib_max = max(lib_max, lib);
        if (!(f & SVT_REC_HAS_PAIR) &&
            (f & (SVT_REC_ALT_STRADDLE | SVT_REC_REF_STRADDLE_A | SVT_REC_REF_STRADDLE_B))) bad |= 2u;
        if (lib >= a.n_libs) { bad |= 4u; continue; }
        if (f & ~SVT_REC_FLAG_MASK) bad |= 8u;
        if ((int32_t)w.x < 0) bad |= 16u;
        if (keeps_pair_entry(w, g, a.libs[lib])) {
            ++o.n[kPairs];
            if ((w.y & 0x8080u) != 0u) o.flags |= kScanWideMapq;
            // half-words of the short layout: one, or an aligned pair (ShortRowWriter below)
            o.n_short += (w.y & 0xffffu) == common ? 1u : (o.n_short & 1u) + 2u;
        }
        uint32_t k[3];
        weight_pairs(w, k);
        o.n[kRefReads] += k[0] ? 1u : 0u;
        o.n[kCandidates] += (k[1] ? 1u : 0u) + (k[2] ? 1u : 0u);
    }
    if (lo == hi) lib_min = 0u;
    o.libs = lib_min | (lib_max << 8);
    a.out[u] = o;
    if (bad) atomicOr(a.err, bad);
}

struct RepackArgs {
    const uint4* csr;
    const uint64_t* lane_src;   // per tile lane: first CSR record of the unit
    const uint32_t* lane_nrec;  // per tile lane: F (0 for padding lanes)
    const TileDesc* tiles;      // in storage order
    const LaneHdr* hdr;         // per tile lane (compact layout: unit geometry + first library)
    const svt_unit* units;
    const LibDesc* libs;
    uint4* tiled;
    uint32_t n_tiles;
    uint32_t multi_lib;         // pair entries carry (lib - lib_min) and 7-bit MAPQs
    uint32_t short_pairs;       // kLayoutShort
    uint32_t common_mq;         // kLayoutShort: the MAPQ pair of the one-half-word entries
};

// dense layout: CSR records -> lane-interleaved rows of 16-byte records
__global__ __launch_bounds__(kBlock) void svt_repack_dense_kernel(const RepackArgs a)
{
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x % kWave;
    const uint32_t tile_idx = blockIdx.x * kWavesPerBlock + wave;
    if (tile_idx >= a.n_tiles) return;
    const TileDesc td = a.tiles[tile_idx];
    const uint64_t src = a.lane_src[td.lane_base + lane];
    const uint32_t nrec = a.lane_nrec[td.lane_base + lane];
    for (uint32_t j = 0; j < td.rows[0]; ++j) {
        const uint4 w = j < nrec ? a.csr[src + j] : make_uint4(0, 0, 0, 0);
        a.tiled[td.base + (uint64_t)j * kWave + lane] = w;
    }
}

// seven 2-byte MAPQ-pair entries per 16-byte row slot of the lane: bytes 0..13; bit k of byte 14 = entry k
// is the first kept one of its fragment; bit k of byte 15 = entry k is a clip candidate (candidate stream)
struct WeightRowWriter {
    uint4* out;       // row 0 of this lane
    uint32_t n = 0;   // entries so far
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    uint32_t stride = kWave;   // slots between consecutive rows of this lane (1: the host packer's unit-major slots)
    __host__ __device__ __forceinline__ void put(const uint32_t mapq_pair, const bool first, const bool clip = false)
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
    __host__ __device__ __forceinline__ void finish(const uint32_t rows)
    {
        uint32_t r = n / 7u;
        if (n % 7u) out[(uint64_t)r++ * stride] = make_uint4(w[0], w[1], w[2], w[3]);
        for (; r < rows; ++r) out[(uint64_t)r * stride] = make_uint4(0, 0, 0, 0);
    }
};

// four 4-byte entries per 16-byte row slot of the lane
struct RowWriter {
    uint4* out;       // row 0 of this lane
    uint32_t n = 0;   // entries so far
    uint4 hold = make_uint4(0, 0, 0, 0);
    __device__ __forceinline__ void put(const uint32_t e)
    {
        switch (n & 3u) {
        case 0: hold.x = e; break;
        case 1: hold.y = e; break;
        case 2: hold.z = e; break;
        default:
            hold.w = e;
            out[(uint64_t)(n >> 2) * kWave] = hold;
            hold = make_uint4(0, 0, 0, 0);
        }
        ++n;
    }
    // flush the open row and zero the rest of the tile's rows (zero entries add +0.0)
    __device__ __forceinline__ void finish(const uint32_t rows)
    {
        uint32_t r = n >> 2;
        if (n & 3u) out[(uint64_t)r++ * kWave] = hold;
        for (; r < rows; ++r) out[(uint64_t)r * kWave] = make_uint4(0, 0, 0, 0);
    }
};

// short layout: eight half-words per 16-byte row slot of the lane
struct ShortRowWriter {
    uint4* out;       // row 0 of this lane
    uint32_t n = 0;   // half-words so far
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    uint32_t stride = kWave;
    __host__ __device__ __forceinline__ void put_half(const uint32_t hw)
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
    __host__ __device__ __forceinline__ void put(const uint32_t lo16, const uint32_t mq, const uint32_t common)
    {
        if (mq == common) {
            put_half(lo16);
        } else {
            if (n & 1u) put_half(0u);          // no-op: wide entries start on a 4-byte boundary
            put_half(lo16 | kWideEntry);
            put_half(mq);
        }
    }
    __host__ __device__ __forceinline__ void finish(const uint32_t rows)
    {
        uint32_t r = n >> 3;
        if (n & 7u) out[(uint64_t)r++ * stride] = make_uint4(w[0], w[1], w[2], w[3]);
        for (; r < rows; ++r) out[(uint64_t)r * stride] = make_uint4(0, 0, 0, 0);
    }
};

// compact layout: CSR records -> the three entry streams of the tile, in record order
__global__ __launch_bounds__(kBlock) void svt_repack_compact_kernel(const RepackArgs a)
{
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x % kWave;
    const uint32_t tile_idx = blockIdx.x * kWavesPerBlock + wave;
    if (tile_idx >= a.n_tiles) return;
    const TileDesc td = a.tiles[tile_idx];
    const uint64_t src = a.lane_src[td.lane_base + lane];
    const uint32_t nrec = a.lane_nrec[td.lane_base + lane];
    const LaneHdr h = a.hdr[td.lane_base + lane];
    UnitGeom g{false, 0, 0.0};
    if (h.unit != kPadUnit) g = unit_geom(a.units[h.unit]);
    const uint32_t lib_min = (h.packed >> 16) & 0xffu;
    uint4* row0 = a.tiled + td.base + lane;
    RowWriter P{row0};
    ShortRowWriter S{row0};
    WeightRowWriter R{row0 + (uint64_t)td.rows[kPairs] * kWave};
    WeightRowWriter X{row0 + (uint64_t)(td.rows[kPairs] + td.rows[kRefReads]) * kWave};
    bool frag_has[3] = {false, false, false};  // did the current fragment already emit an entry for this tally?
    for (uint32_t j = 0; j < nrec; ++j) {
        const uint4 w = a.csr[src + j];
        if (!(w.w & SVT_REC_CONTINUATION)) frag_has[0] = frag_has[1] = frag_has[2] = false;
        const uint32_t lib_idx = SVT_REC_LIB(w.w);
        const LibDesc lib = a.libs[lib_idx];
        if (keeps_pair_entry(w, g, lib)) {
            const uint32_t lo16 = (w.w & 7u) | (pair_code(w.x, g, lib) << 3);
            const uint32_t mq_a = w.y & 0xffu, mq_b = (w.y >> 8) & 0xffu;
            if (a.short_pairs) S.put(lo16, w.y & 0xffffu, a.common_mq);
            else P.put(a.multi_lib ? (lo16 | ((lib_idx - lib_min) << 16) | (mq_a << 18) | (mq_b << 25))
                                   : (lo16 | (mq_a << 16) | (mq_b << 24)));
        }
        uint32_t k[3];
        weight_pairs(w, k);
        if (k[0]) {
            R.put(k[0], !frag_has[0]);
            frag_has[0] = true;
        }
#pragma unroll
        for (int s = 1; s < 3; ++s)
            if (k[s]) {
                X.put(k[s], !frag_has[s], s == 2);
                frag_has[s] = true;
            }
    }
    if (a.short_pairs) S.finish(td.rows[kPairs]);
    else P.finish(td.rows[kPairs]);
    R.finish(td.rows[kRefReads]);
    X.finish(td.rows[kCandidates]);
}


}  // namespace svt

#endif  // SVT_PREPARE_KERNELS_H
