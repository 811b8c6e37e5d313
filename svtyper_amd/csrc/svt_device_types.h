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

// packed evidence: the pair stream is counted in half-words (2-byte entries, 4-byte wide ones), eight per 16-byte slot
constexpr uint32_t kHalfwordsPerRow = 8u;

// record-contract violations (include/svtyper_hip.h: svt_record), as the streaming pass ORs them into its error word
// and as the host encoder of packed evidence reports them
constexpr uint32_t kErrStraddleNoPair = 2u, kErrLibIndex = 4u, kErrReservedBits = 8u, kErrNegativeSpan = 16u;

// device layout of a batch's evidence
enum Layout : int {     // (0..2 were round 1's tiled layouts)
    kLayoutStream = 3,   // the caller's CSR as it is, streamed through per-wave LDS rings (svt_stream_kernel.h)
    kLayoutPacked = 4    // packed evidence (svt_packed_evidence) as uploaded, streamed the same way (svt_packed_kernel.h)
};
// one bin of a library's insert-size tables: thr = largest h2 for which p_concordant still holds
// (svt_host_tables.h), hist = the Counter value -- both replaced by their rank among the library's values, which
// is all `hist[a] <= thr[b]` needs.  Every library owns n_bins + 1 of them; the last one is the out-of-range
// sentinel {-1, rank of 0}.
struct Bin {
    int32_t thr;
    uint32_t hist;
};

// LDS layout of the packed-evidence kernel, in bytes from the start of the workgroup's LDS (the kernel has
// no static LDS, so the dynamic segment starts at 0 -- checked at run time).  The first regions have fixed
// addresses; the entries are consumed with these as immediates.  (svt_stream_kernel.h has its own: kS*.)
constexpr uint32_t kLdsPm = 0;                       // double[256]      prob_mapq
constexpr uint32_t kLdsWtab = kLdsPm + 256 * 8;      // PairWeights[32]  paired-end decision table
// the entries read the decision table column-wise: w_alt[32] then w_ref[32], 8-byte rows.  Rows that differ
// only in p_concordant then sit in different LDS banks (with 16-byte {w_alt, w_ref} rows every (p_concordant, is_DEL)
// variant of a straddle pattern shares its four banks and the lanes of a wave serialise on them)
constexpr uint32_t kLdsWcol = kLdsWtab + 32 * 16;    // double[32] w_alt, double[32] w_ref
constexpr uint32_t kLdsWcolC = kLdsWcol + 2 * 32 * 8;   // the same two columns times pmA * pmB of the batch's common MAPQ pair 
constexpr uint32_t kWcolRef = 32 * 8;                // byte distance from a w_alt entry to its w_ref entry
constexpr uint32_t kLdsBins = kLdsWcolC + 2 * 32 * 8;   // int32 thr[total_bins], uint32 hist[total_bins], then log10, then the rings

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

// Library window of one workgroup: the descriptors [lib_lo, lib_lo + lib_cnt) and the histogram/threshold
// bins [bin_lo, bin_lo + bin_cnt) are the only ones its records can reference, so only they are staged in LDS
// (kMultiLds).  A window normally holds the 1..3 libraries of one sample (svt_unit.libs).
struct WgDesc {
    uint32_t lib_lo, lib_cnt, bin_lo, bin_cnt;
};


}  // namespace svt

#endif  // SVT_DEVICE_TYPES_H
