// svt_common.h -- constants, error reporting and call-checking macros
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_COMMON_H
#define SVT_COMMON_H

#include "svt_error.h"

namespace svt {


// ------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------
constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kWavesPerBlock = 4;    // 256-thread workgroups
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr uint32_t kChunkUnits = SVT_CHUNK;  // sort window (units)
constexpr uint32_t kPadUnit = 0xFFFFFFFFu;
constexpr uint32_t kMaxLdsTableBytes = 64 * 1024;  // hist+thr budget before falling back to HBM/L2 tables
constexpr uint32_t kMaxL10Lds = 4096;              // log10 table entries kept in LDS (32 KiB)
constexpr uint32_t kCodeBits = 13;      // compact layout: bits of a pair entry's table code (svt_prepare_kernels.h)
constexpr uint32_t kMaxShortBins = 2047;  // short layout: 2 * n_bins must fit the 12-bit code of a half-word entry
constexpr uint32_t kMaxCompactBins = ((1u << kCodeBits) - 1) / 2;   // code <= 2 * n_bins must fit
constexpr uint32_t kMaxCompactLibSpan = 4;                          // libraries per unit (2-bit local index)
constexpr uint32_t kTailPadRows = 16;   // look-ahead loads may run this far past a tile (>= 2 * group)


#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return fail(SVT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)
#define SVT_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != SVT_OK) return _rc; \
    } while (0)


}  // namespace svt

#endif  // SVT_COMMON_H
