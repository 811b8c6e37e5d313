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
    uint32_t tab_off;     // offset of this library's bins inside hist[] / thr[] (each library
                          // owns n_bins + 1 entries; the last one is the out-of-range sentinel)
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

// One 64-unit tile.  Dense layout: rows_a rows of 16-byte records at base_a (rows_b == 0).
// Compact layout: rows_a rows of pair entries at base_a, rows_b rows of weight entries at base_b
// (each 16-byte row slot of a lane holds four consecutive 4-byte entries).
struct TileDesc {         // 32 B, stored in dispatch (longest-first) order
    uint64_t base_a;
    uint64_t base_b;
    uint32_t rows_a;
    uint32_t rows_b;
    uint32_t lane_base;   // first LaneHdr of the tile
    uint32_t pad;
};

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
    const uint32_t* hist;      // total_bins (sentinels included)
    const int32_t* thr;        // total_bins
    const PairWeights* wtab;   // 32
    uint32_t n_l10;
    uint32_t n_libs;
    uint32_t total_bins;
    uint32_t n_tiles;
    uint32_t l10_in_lds;
    uint32_t lds_libs;         // LDS capacity in library descriptors (largest window)
    uint32_t lds_bins;         // LDS capacity in histogram bins (largest window)
    uint32_t pad0;
    uint64_t n_units;
    svt_result* out;           // [n_units]
    LibDesc lib0;              // copy of libs[0] (kSingleLds)
    GtConsts c;
};


}  // namespace svt

#endif  // SVT_DEVICE_TYPES_H
