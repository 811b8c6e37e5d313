// svt_device_types.h -- structures shared by the host code and the kernels
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_DEVICE_TYPES_H
#define SVT_DEVICE_TYPES_H

#include "svt_common.h"

namespace svt {

// ------------------------------------------------------------------------------------------
// device-side structures
// ------------------------------------------------------------------------------------------
struct LibDesc {          // 32 B, one per library
    uint32_t tab_off;     // offset of this library's bins inside bins[] (each library owns
                          // n_bins + 1 entries; the last one is the out-of-range sentinel)
    int32_t key_min;
    uint32_t n_bins;
    uint32_t pad;
    double v_nondel;      // lib.mean + lib.sd * 3   (parsers.py:873-875)
    double sd2;           // 2 * lib.sd              (classic.py:339)
};

struct LaneHdr {          // 16 B, one per tile lane
    int32_t var_length;
    int32_t pos_delta;
    uint32_t unit;        // original unit index, kPadUnit for padding lanes
    uint32_t packed;      // svtype | flags << 8 | first library of the unit << 16
};

// One 64-unit tile: `rows[k]` rows of stream k, stored back to back from `base` (in 16-byte row
// slots; row j holds the j-th 16 bytes of each of the 64 lanes).  Dense layout: one stream of
// canonical 16-byte records (rows[0] = longest unit).  Compact layout: three streams of entries --
// pair entries (4 bytes, four per row slot), reference-read weight entries and split/clip candidate
// weight entries (2 bytes, seven per row slot) (svt_prepare_kernels.h has the formats).
constexpr int kStreams = 3;
enum Stream : int { kPairs = 0, kRefReads = 1, kCandidates = 2 };
// entries per 16-byte row slot: 4-byte pair entries; 2-byte reference-read and candidate entries
// (seven MAPQ pairs in bytes 0..13, their flag bits in bytes 14..15)
constexpr uint32_t kEntriesPerRow[kStreams] = {4u, 7u, 7u};
// kLayoutShort: the pair stream is counted in half-words (2-byte short entries, 4-byte wide ones), eight per slot
constexpr uint32_t kHalfwordsPerRow = 8u;

// device layout of a batch's evidence
enum Layout : int {
    kLayoutDense = 0,    // canonical 16-byte records
    kLayoutCompact = 1,  // three entry streams, 4-byte pair entries
    kLayoutShort = 2,    // three entry streams, 2-byte pair entries for the batch's most common MAPQ pair
    kLayoutStream = 3,   // the caller's CSR as it is, streamed through per-wave LDS rings (svt_stream_kernel.h)
    kLayoutPacked = 4    // packed evidence (svt_packed_evidence) as uploaded, streamed the same way (svt_packed_kernel.h)
};
struct TileDesc {         // 32 B, stored in dispatch (longest-first) order
    uint64_t base;
    uint32_t rows[kStreams];
    uint32_t lane_base;   // first LaneHdr of the tile
    uint32_t pad[2];
};

// one bin of a library's insert-size tables: thr = largest h2 for which p_concordant still holds
// (svt_host_tables.h), hist = the Counter value.  Every library owns n_bins + 1 of them; the last one
// is the out-of-range sentinel {-1, 0}.
struct Bin {
    int32_t thr;
    uint32_t hist;
};

// LDS layout of the genotype kernel, in bytes from the start of the workgroup's LDS (the kernel has
// no static LDS, so the dynamic segment starts at 0 -- checked at run time).  The first three
// regions have fixed addresses; the compact entries are consumed with these as immediates.
constexpr uint32_t kLdsPm = 0;                       // double[256]      prob_mapq
constexpr uint32_t kLdsWtab = kLdsPm + 256 * 8;      // PairWeights[32]  paired-end decision table
// the compact layouts read the decision table column-wise: w_alt[32] then w_ref[32], 8-byte rows.  Rows that differ
// only in p_concordant then sit in different LDS banks (with 16-byte {w_alt, w_ref} rows every (p_concordant, is_DEL)
// variant of a straddle pattern shares its four banks and the lanes of a wave serialise on them)
constexpr uint32_t kLdsWcol = kLdsWtab + 32 * 16;    // double[32] w_alt, double[32] w_ref
constexpr uint32_t kLdsWcolC = kLdsWcol + 2 * 32 * 8;   // the same two columns times pmA * pmB of the batch's common MAPQ pair (short layout)
constexpr uint32_t kWcolRef = 32 * 8;                // byte distance from a w_alt entry to its w_ref entry
constexpr uint32_t kLdsBins = kLdsWcolC + 2 * 32 * 8;   // Bin[lds_bins], then LibDesc[lds_libs], then log10

struct GtConsts {
    double lgp[2][3];     // [is_dup][genotype] log(p)/log(10)      (statistics.py:33-35)
    double lg1p[2][3];    // [is_dup][genotype] log(1-p)/log(10)
    double ln10;          // log(10.0)
    double x_uflow;       // smallest x with libm pow(10.0, x) > 0
    double split_weight;
    double disc_weight;
};

// Paired-end decision table (classic.py:359-405), 32 entries of {w_alt, w_ref}:
//   index = alt_straddle | ref_straddle_A << 1 | ref_straddle_B << 2 | p_concordant << 3 | is_DEL << 4
//   alt_span += (pmA * pmB) * w_alt      w_alt in {0, 1}
//   ref_span += (pmA * pmB) * w_ref      w_ref in {0, 0.5, 1}     ((A + B) * p / 2)
// Multiplying a finite non-negative binary64 by 0, 0.5 or 1 is exact, so this is the reference's
// arithmetic with the branch structure moved into a lookup.
struct PairWeights { double w_alt, w_ref; };

enum LibMode : int {
    kSingleLds = 0,  // one library, descriptor in SGPRs, tables in LDS, 32-bit index math
    kMultiLds = 1,   // several libraries, descriptors + tables in LDS, 32-bit index math
    kGeneral = 2     // any geometry: 64-bit index math, exact float Counter key, tables in HBM/L2
};

// Library window of one workgroup (its 4 tiles): the descriptors [lib_lo, lib_lo + lib_cnt) and the
// histogram/threshold bins [bin_lo, bin_lo + bin_cnt) are the only ones its records can reference, so
// only they are staged in LDS (kMultiLds).  Units are sorted by library first, so a window normally
// holds the 1..3 libraries of one sample.
struct WgDesc {
    uint32_t lib_lo, lib_cnt, bin_lo, bin_cnt;
};

struct KernelArgs {
    const uint4* tiled;
    const TileDesc* tiles;
    const WgDesc* wg;          // one per workgroup (kMultiLds)
    const LaneHdr* hdr;
    const double* pm;          // 256
    const double* l10;         // n_l10
    const LibDesc* libs;       // n_libs
    const Bin* bins;           // total_bins (sentinels included)
    const PairWeights* wtab;   // 32
    uint32_t n_l10;
    uint32_t n_libs;
    uint32_t total_bins;
    uint32_t n_tiles;
    uint32_t l10_in_lds;
    uint32_t lds_libs;         // LDS capacity in library descriptors (largest window)
    uint32_t lds_bins;         // LDS capacity in histogram bins (largest window)
    uint32_t common_mq;        // kLayoutShort: mapq_a | mapq_b << 8 of the short pair entries
    uint64_t n_units;
    svt_result* out;           // [n_units]
    LibDesc lib0;              // copy of libs[0] (kSingleLds)
    GtConsts c;
};


}  // namespace svt

#endif  // SVT_DEVICE_TYPES_H
