// svt_window_scan_kernel.h -- library windows without hints: which libraries does each unit's evidence name?
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
//
// A batch of several libraries runs fastest when a workgroup stages only the histograms its units need
// (svt_stream_kernel.h, kMultiLds).  The producer can say which those are (svt_unit.libs: the libraries of the
// unit's sample, parsers.py:432-447); when it does not, svt_batch_create asks the records themselves, once, right
// after the upload: one streaming read of the library byte of every record (the lanes of a wave take consecutive
// records, so the loads cover whole lines) reduced to [first, last] per unit.  The result has the encoding of the
// hint, SVT_UNIT_LIBS(first, count); the host groups the units by it exactly as it groups hinted units, and the pass
// checks every record against its window like it checks a hinted one.
#ifndef SVT_WINDOW_SCAN_KERNEL_H
#define SVT_WINDOW_SCAN_KERNEL_H

#include "svt_common.h"
#include "svt_device_types.h"

namespace svt {

constexpr uint32_t kScanBlock = 256;

__global__ __launch_bounds__(kScanBlock) void svt_window_scan_kernel(const uint4* __restrict__ records, const uint64_t* __restrict__ rec_offset,
                                                                     const uint32_t n_units, uint32_t* __restrict__ out)
{
    const uint32_t lane = threadIdx.x % kWave;
    const uint32_t n_waves = gridDim.x * (kScanBlock / kWave);
    const uint32_t* flags = reinterpret_cast<const uint32_t*>(records) + 3;   // dword 3 of record r: flags[4 r]
    for (uint32_t u = (blockIdx.x * kScanBlock + threadIdx.x) / kWave; u < n_units; u += n_waves) {
        const uint64_t r0 = rec_offset[u], r1 = rec_offset[u + 1];
        uint32_t lo = 0xffffu, hi = 0u;
        uint64_t r = r0 + lane;
        for (; r + (uint64_t)kWave < r1; r += 2 * (uint64_t)kWave) {      // two loads in flight per lane
            const uint32_t a = SVT_REC_LIB(flags[4 * r]), b = SVT_REC_LIB(flags[4 * (r + kWave)]);
            lo = min(lo, min(a, b));
            hi = max(hi, max(a, b));
        }
        if (r < r1) {
            const uint32_t a = SVT_REC_LIB(flags[4 * r]);
            lo = min(lo, a);
            hi = max(hi, a);
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, kWave));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, kWave));
        }
        // a unit without records names no library: any window will do (the first library's)
        // (0 = no window: more than 255 libraries apart -- the hint's count field is eight bits)
        if (lane == 0) out[u] = r1 <= r0 ? SVT_UNIT_LIBS(0u, 1u) : hi - lo + 1u <= 255u ? SVT_UNIT_LIBS(lo, hi - lo + 1u) : 0u;
    }
}

}  // namespace svt

#endif  // SVT_WINDOW_SCAN_KERNEL_H
