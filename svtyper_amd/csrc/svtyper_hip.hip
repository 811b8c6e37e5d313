// svtyper_hip.hip -- MI355X (gfx950 / CDNA4) implementation of the SVTyper likelihood
// hot path behind the C ABI of include/svtyper_hip.h.
//
// Reference lines restated on the device (paths relative to the reference checkout):
//   per-fragment tallies        svtyper/classic.py:296-408, svtyper/singlesample.py:246-353
//   prob_mapq                   svtyper/utils.py:74-75          (256-entry LUT in LDS)
//   p_concordant                svtyper/parsers.py:861-882      (histogram + threshold table in LDS)
//   zeroing rules               svtyper/classic.py:425-435
//   QR/QA + counts              svtyper/classic.py:442-444,455-465
//   log_choose / bayes_gt       svtyper/statistics.py:9-37      (log10 table in LDS, same O(k) loop)
//   GT/GQ/SQ decision           svtyper/classic.py:446,473-495
//
// Design (DESIGN.md has the long form):
//   * one (breakpoint, sample) unit per lane: the five tallies are sequential binary64 sums in record order,
//     exactly as CPython evaluates them, so the truncated integer counts are bit-exact.  No FMA contraction
//     (-ffp-contract=off).
//   * nothing is re-tiled or re-encoded on the way: the caller's CSR (rec_offset / unit headers / 16-byte records)
//     goes to HBM as it is and ONE kernel takes it to the result records (svt_stream_kernel.h).  A workgroup of 256
//     consecutive units sorts them by length, every wave streams its 64 units' records through an LDS ring filled
//     by LDS-DMA (one whole 128-byte line per unit and step, svt_ring_engine.h), the lanes consume them in order.
//   * a producer behind PCIe can hand over packed evidence instead (svt_pack_evidence: three sparse streams of
//     2/4-byte entries per unit, ~3 bytes per record; svt_entry_formats.h has the formats, svt_packed_kernel.h the
//     pass): entries that could only add +0.0 are dropped, which the reference's sums cannot observe.
//   * all look-up tables (prob_mapq, insert-size histogram + p_concordant thresholds as 16-bit ranks, log10,
//     paired-end decision weights) are built on the host with the same libm CPython uses and staged in LDS per
//     workgroup; several libraries: per-sample library windows (svt_unit.libs), else tables through L2.
//   * the one-shot entry points overlap upload, pass and download by unit ranges on three streams (run_pipelined).
//   * HBM-bound byte/integer/fp64 streaming: no MFMA anywhere (nothing here is a contraction).
//
// There is no CPU fallback in this file.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/svtyper_hip.h"

#ifndef SVT_WINDOW_TILES
#define SVT_WINDOW_TILES 2   // library-window mode: tiles per wave of large launches (1 = always one)
#endif
#ifndef SVT_STREAM_R
#define SVT_STREAM_R 1   // streaming kernel: 64-unit tiles per wave (a workgroup sorts 256 * R consecutive units)
#endif

#include "svt_common.h"
#include "svt_device_types.h"
#include "svt_unit_math.h"
#include "svt_entry_formats.h"
#include "svt_pack.h"
#include "svt_stream_kernel.h"
#include "svt_packed_kernel.h"
#include "svt_coop_kernel.h"
#include "svt_split_kernel.h"
#include "svt_window_scan_kernel.h"
#include "svt_geometry_kernel.h"
#include "svt_bayes_kernel.h"
#include "svt_host_tables.h"
#include "svt_host_transfer.h"

using namespace svt;

// the one-tile-per-wave instantiations of the streaming kernel live in svt_small_kernels.hip (their own scheduling strategy)
#if SVT_STREAM_R == 1 && !defined(SVT_NO_SMALL_TU)
namespace svt {
extern template __global__ void svt_stream_kernel<false, kSingleLds, 1>(const StreamArgs);
extern template __global__ void svt_stream_kernel<true, kSingleLds, 1>(const StreamArgs);
extern template __global__ void svt_stream_kernel<false, kMultiLds, 1>(const StreamArgs);
extern template __global__ void svt_stream_kernel<true, kMultiLds, 1>(const StreamArgs);
extern template __global__ void svt_split_kernel<false, kSingleLds, 2>(const StreamArgs);
extern template __global__ void svt_split_kernel<true, kSingleLds, 2>(const StreamArgs);
extern template __global__ void svt_split_kernel<false, kSingleLds, 4>(const StreamArgs);
extern template __global__ void svt_split_kernel<true, kSingleLds, 4>(const StreamArgs);
extern template __global__ void svt_split_kernel<false, kMultiLds, 2>(const StreamArgs);
extern template __global__ void svt_split_kernel<false, kMultiLds, 4>(const StreamArgs);
extern template __global__ void svt_split_kernel<true, kMultiLds, 4>(const StreamArgs);
}  // namespace svt
#endif

// ------------------------------------------------------------------------------------------
// batch object
// ------------------------------------------------------------------------------------------
constexpr unsigned kKnownFlags = SVT_FLAG_SSO_ASSOCIATION | SVT_FLAG_GENERAL_TABLES | SVT_FLAG_RESULT96;

struct svt_batch {
    int device = 0;
    unsigned flags = 0;
    hipStream_t stream = nullptr;
    uint64_t n_units = 0, n_records = 0;
    int mode = kSingleLds;
    int layout = kLayoutStream;   // svt_device_types.h: Layout
    size_t lds_bytes = 0;
    bool have_results = false;
    // device buffers; the big ones come from (and go back to) g_pool, the small ones from g_handles
    uint64_t cap_out = 0;
    double* d_pm = nullptr;
    double* d_l10 = nullptr;
    LibDesc* d_libs = nullptr;
    Bin* d_bins = nullptr;
    PairWeights* d_wtab = nullptr;
    svt_result* d_out = nullptr;
    svt_result* out_dev = nullptr;   // where the pass writes: d_out, or the buffer of svt_batch_bind_device_results
    // kLayoutStream: the canonical CSR as it is (svt_stream_kernel.h); d_records / d_off / d_units come from g_pool
    void* d_records = nullptr;
    uint64_t* d_off = nullptr;
    svt_unit* d_units = nullptr;
    uint64_t cap_records = 0, cap_off = 0, cap_units = 0;
    uint32_t* d_err = nullptr;
    uint32_t* d_perm = nullptr;      // kMultiLds (library windows): units grouped by window, chunk list, windows
    uint2* d_chunks = nullptr;
    WgDesc* d_windows = nullptr;
    uint64_t cap_perm = 0;
    uint32_t n_chunks = 0;
    uint32_t n_chunks_one = 0;       // ... of which the first ones have windows of ONE library
    bool split_window_kinds = false; // the pass is two launches: one-library windows, then the others (a kernel per kind)
    int window_tiles = 1;            // kMultiLds: 64-unit tiles per wave (chunks hold up to 256 * window_tiles units)
    uint64_t bound_slots = 0;        // svt_batch_bind_device_results: result records the caller's buffer holds (0 = the library's own buffer)
    uint64_t out_slots = 0;          // records in the device result buffer after a pass: n_units, or (SVT_FLAG_RESULT96) the slots of
                                     // the pass's workgroups -- tagged records in the kernel's order, padding included
    bool records_resident = true;    // false: create_stream left the record upload to its caller (pipelined one-shot)
    int wgs_per_cu = 3;              // workgroups per CU the pass's kernel was budgeted for (registers -> LDS per workgroup)
    uint32_t resident_wgs = 0;       // workgroups of the pass's kernel the device holds at once (registers, LDS, CUs); 0 = unknown
    uint64_t one_tile_round_units = 0;   // units ONE round of the one-tile-per-wave kernel's resident workgroups holds (one library); 0 = unknown
    // the cooperative kernel for launches of less than one round (svt_coop_kernel.h); 0 bytes = not for this batch
    size_t coop_lds_bytes = 0;
    uint32_t coop_region = 0, coop_l10_where = kL10Global, coop_lds_l10 = 0, coop_l10_entries = 0;
    uint32_t coop_resident = 0;      // workgroups of it the device holds at once
    // K lanes per unit (svt_split_kernel.h): the same for its region
    size_t split_lds_bytes = 0;
    uint32_t split_region = 0, split_l10_where = kL10Global, split_lds_l10 = 0, split_l10_entries = 0;
    StreamArgs sargs{};
    // kLayoutPacked: packed evidence as uploaded (svt_packed_kernel.h); d_records holds the slots, d_soff the 3n+1 offsets
    uint32_t* d_soff = nullptr;
    uint64_t cap_soff = 0, n_slots = 0;
    PackedArgs pargs{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

#include "svt_batch_state.h"
#include "svt_batch_create.h"
#include "svt_batch_packed.h"
#include "svt_batch_oneshot.h"

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

#include "svt_entry_batch.h"
#include "svt_entry_seams.h"
#include "svt_entry_packed.h"

void* svt_batch_stream(svt_batch* b) { return b ? (void*)b->stream : nullptr; }

void svt_batch_destroy(svt_batch* b) { free_batch(b); }

#include "svt_entry_debug.h"
#include "svt_entry_oneshot.h"

}  // extern "C"
