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
constexpr uint32_t kPadUnit = 0xFFFFFFFFu;
constexpr uint32_t kMaxShortBins = 2047;  // packed evidence: 2 * n_bins must fit the 12-bit code of a half-word pair entry


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
