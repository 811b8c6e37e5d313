// svt_small_kernels.hip -- the ONE-TILE-PER-WAVE instantiations of svt_stream_kernel, a translation unit of their own
// Part of libsvtyper_hip.so; the host code in svtyper_hip.hip launches these through `extern template` declarations.
//
// One tile per wave is what launches of less than a round of resident workgroups take (svtyper_hip.hip: tiles_per_wave): a
// SIMD then holds one or two waves, and a wave that has its SIMD nearly to itself issues an instruction every ~8 cycles when
// the next one depends on the last -- every ~4-5 when it does not (tools/lat_probe.hip, profiles/r05_small_launch_probes.txt).
// The default scheduling strategy orders instructions for occupancy (few live registers, dependent pairs back to back);
// `-mllvm -amdgpu-sched-strategy=max-ilp` interleaves independent chains.  In-process A/B over the same memory: 20 k units
// 0.0427 -> 0.0356 ms, 125 k units 0.0557 -> 0.0523; at 1 M units (two tiles per wave, four workgroups per CU) it costs 1 %,
// which is why it is a compile flag of THIS file only (svtyper_amd/csrc/Makefile) and not of the library.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/svtyper_hip.h"

#include "svt_common.h"
#include "svt_device_types.h"
#include "svt_unit_math.h"
#include "svt_stream_kernel.h"
#include "svt_split_kernel.h"

namespace svt {
template __global__ void svt_stream_kernel<false, kSingleLds, 1>(const StreamArgs);
template __global__ void svt_stream_kernel<true, kSingleLds, 1>(const StreamArgs);
template __global__ void svt_stream_kernel<false, kMultiLds, 1>(const StreamArgs);
template __global__ void svt_stream_kernel<true, kMultiLds, 1>(const StreamArgs);
// K lanes per unit (svt_split_kernel.h)
template __global__ void svt_split_kernel<false, kSingleLds, 2>(const StreamArgs);
template __global__ void svt_split_kernel<true, kSingleLds, 2>(const StreamArgs);
template __global__ void svt_split_kernel<false, kSingleLds, 4>(const StreamArgs);
template __global__ void svt_split_kernel<true, kSingleLds, 4>(const StreamArgs);
template __global__ void svt_split_kernel<false, kMultiLds, 2>(const StreamArgs);
template __global__ void svt_split_kernel<false, kMultiLds, 4>(const StreamArgs);
template __global__ void svt_split_kernel<true, kMultiLds, 4>(const StreamArgs);
}  // namespace svt
