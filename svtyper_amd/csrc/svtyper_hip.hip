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

void free_batch(svt_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    // nothing of this batch may still be in flight when its buffers go back to the pool for the next one to take
    // (a create that failed half way has copies enqueued; a caller may destroy right after an asynchronous pass)
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    auto F = [](void* p) { g_handles.put_small(p); };
    g_pool.put(b->device, b->d_out, b->cap_out);
    g_pool.put(b->device, b->d_records, b->cap_records);
    g_pool.put(b->device, b->d_off, b->cap_off);
    g_pool.put(b->device, b->d_units, b->cap_units);
    g_pool.put(b->device, b->d_soff, b->cap_soff);
    g_pool.put(b->device, b->d_perm, b->cap_perm);
    F(b->d_chunks); F(b->d_windows);
    F(b->d_err);
    F(b->d_pm); F(b->d_l10); F(b->d_libs);
    F(b->d_bins); F(b->d_wtab);
    g_handles.put_event(b->ev0, true);
    g_handles.put_event(b->ev1, true);
    g_handles.put_stream(b->stream);   // (idle: synchronised above)
    delete b;
}

// device scratch that only lives during svt_batch_create
struct DevScratch {
    void* p = nullptr;
    ~DevScratch() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        HIP_TRY(hipMalloc(&p, std::max<size_t>(bytes, 16)));
        return SVT_OK;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

template <typename T>
int upload(T** dptr, const std::vector<T>& v, Stager& st)
{
    void* p = nullptr;
    const uint64_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    if (bytes <= (1u << 20)) SVT_TRY(g_handles.get_small(bytes, &p));     // free_batch hands these back (put_small)
    else HIP_TRY(hipMalloc(&p, bytes));
    *dptr = static_cast<T*>(p);
    return st.copy(*dptr, v.data(), v.size() * sizeof(T));
}

template <typename T>
int upload(DevScratch& d, const std::vector<T>& v, Stager& st)
{
    SVT_TRY(d.alloc(v.size() * sizeof(T)));
    return st.copy(d.p, v.data(), v.size() * sizeof(T));
}

template <bool SSO>
const void* stream_kernel_for(int mode, int tiles)
{
    if (mode == kSingleLds && tiles == 2) return reinterpret_cast<const void*>(&svt_stream_kernel<SSO, kSingleLds, 2>);
    if (mode == kMultiLds && tiles == 2) return reinterpret_cast<const void*>(&svt_stream_kernel<SSO, kMultiLds, 2>);
    return mode == kSingleLds  ? reinterpret_cast<const void*>(&svt_stream_kernel<SSO, kSingleLds, SVT_STREAM_R>)
           : mode == kMultiLds ? reinterpret_cast<const void*>(&svt_stream_kernel<SSO, kMultiLds, SVT_STREAM_R>)
                               : reinterpret_cast<const void*>(&svt_stream_kernel<SSO, kGeneral, SVT_STREAM_R>);
}

// library windows, classic association, two tiles per wave: a kernel per kind of window (1 = one library, 2 = several)
const void* window_kernel_of_kind(int kind)
{
    return kind == 1 ? reinterpret_cast<const void*>(&svt_stream_kernel<false, kMultiLds, 2, 1>)
                     : reinterpret_cast<const void*>(&svt_stream_kernel<false, kMultiLds, 2, 2>);
}

const void* stream_kernel_of(const svt_batch* b, int tiles = SVT_STREAM_R)
{
    return (b->flags & SVT_FLAG_SSO_ASSOCIATION) ? stream_kernel_for<true>(b->mode, tiles) : stream_kernel_for<false>(b->mode, tiles);
}

const void* coop_kernel_of(const svt_batch* b)
{
    return (b->flags & SVT_FLAG_SSO_ASSOCIATION) ? reinterpret_cast<const void*>(&svt_coop_kernel<true, kSingleLds>)
                                                  : reinterpret_cast<const void*>(&svt_coop_kernel<false, kSingleLds>);
}

const void* split_kernel_of(const svt_batch* b, int lanes)
{
    const bool sso = (b->flags & SVT_FLAG_SSO_ASSOCIATION) != 0;
    if (b->mode == kMultiLds) {   // (singlesample association: four lanes only -- split_lanes_for never asks for two)
        if (lanes == 2) return reinterpret_cast<const void*>(&svt_split_kernel<false, kMultiLds, 2>);
        return sso ? reinterpret_cast<const void*>(&svt_split_kernel<true, kMultiLds, 4>) : reinterpret_cast<const void*>(&svt_split_kernel<false, kMultiLds, 4>);
    }
    if (lanes == 2) return sso ? reinterpret_cast<const void*>(&svt_split_kernel<true, kSingleLds, 2>) : reinterpret_cast<const void*>(&svt_split_kernel<false, kSingleLds, 2>);
    return sso ? reinterpret_cast<const void*>(&svt_split_kernel<true, kSingleLds, 4>) : reinterpret_cast<const void*>(&svt_split_kernel<false, kSingleLds, 4>);
}

// 64-unit tiles per wave for a launch over `units` units.  One tile per wave leaves a third of a workgroup's
// wave-time waiting for the wave that holds its longest units; two tiles in snake order even that out (DESIGN.md
// 3.1) but make a workgroup run longer, which pays once the one-tile launch would need more than one round of
// resident workgroups: measured -17 % at 250 k units, +-1 % at 500 k, -9 % at 1 M, -6 % at 2 M; +10 % at exactly
// one round (196 608), no difference below.  One library only (the other modes are register-bound).
// (This round the one-tile kernels are compiled for instruction-level parallelism -- svt_small_kernels.hip: 149 VGPRs, three
// workgroups per CU --, so one round of them is 768 workgroups = 196 608 units: a launch beyond it, which would take a second
// round of one-tile workgroups, takes two tiles per wave: 200 k units 0.106 -> 0.087 ms.)
constexpr uint64_t kTwoTilesMinUnits = 768ull * kBlock * 9 / 8;   // library windows: a little more than the chip's resident workgroups hold
int tiles_per_wave(const svt_batch* b, uint64_t units)
{
#if SVT_STREAM_R == 1 && !defined(SVT_STREAM_ONE_TILE)
    if (b->mode == kSingleLds && units > (b->one_tile_round_units ? b->one_tile_round_units : kTwoTilesMinUnits)) return 2;
#endif
    (void)b; (void)units;
    return SVT_STREAM_R;
}

// one launch of the streaming kernel over units [a.unit_begin, a.unit_end) (not the library-window mode)
#ifndef SVT_L10_THROUGH_RING
#define SVT_L10_THROUGH_RING 0   // (in-process A/B, one-library pass with four workgroups per CU: through L2 0.3181 ms, head through the ring 0.3281)
//  a log10 table that does not fit beside the tables: 1 = its head through the wave's ring before each epilogue, 0 = all of it through L2
#endif

// result slots (SVT_FLAG_RESULT96: whole workgroups of tagged records) a launch over `units` units of this batch writes;
// not the library-window mode, whose launch covers b->n_chunks window chunks
// compute units of a device (the chip's resident workgroups = workgroups per CU x this)
inline uint32_t cu_count(int device)
{
    static std::mutex lock;
    static std::vector<int> known;
    std::lock_guard<std::mutex> g(lock);
    if ((size_t)device >= known.size()) known.resize((size_t)device + 1, 0);
    if (known[(size_t)device] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
        known[(size_t)device] = v;
    }
    return (uint32_t)known[(size_t)device];
}

// How a launch over `units` units is cut into workgroups.  The pass is memory-bound and its workgroups run in near lockstep:
// the chip holds `resident_wgs` of them, a launch takes as many ROUNDS of that as its workgroups need, and a last round of a few
// workgroups costs a third of a full one whatever it holds (measured over the same buffers, 512-unit workgroups, 1 024 resident:
// 1 024 workgroups 0.165 ms, 1 094 -> 0.211; 2 048 -> 0.322, 2 090 -> 0.359; profiles/r04_wg_rounds.txt).  So the units of
// a launch of more than one round are dealt out as EQUAL workgroups that fill whole rounds -- 1 M units: 2 045 workgroups of
// 489 instead of 1 954 of 512 -- unless that would leave more than a quarter of a workgroup's lanes (and tagged result slots) empty.
#ifndef SVT_WG_BALANCE
#define SVT_WG_BALANCE 1
#endif
#ifndef SVT_WG_MIN_FILL
#define SVT_WG_MIN_FILL 50   // per cent: the emptiest workgroup the rule may make (more than one round: never below 50)
#endif
struct WgPlan { int tiles; uint32_t per_wg, n_wg; bool coop; int split; };   // split: lanes per unit of svt_split_kernel (0 = not that kernel)
static std::atomic<int> g_wg_balance{SVT_WG_BALANCE && !std::getenv("SVT_NO_WG_BALANCE") ? SVT_WG_MIN_FILL : 0};   // (svt_debug_wg_balance: measurements)
extern "C" int svt_debug_wg_balance(int min_fill_percent) { return g_wg_balance.exchange(std::max(0, std::min(100, min_fill_percent))); }
inline uint32_t balanced_units_per_wg(uint64_t units, uint64_t n_min, uint32_t full, uint32_t resident)
{
    const int min_fill = g_wg_balance.load(std::memory_order_relaxed);     // per cent of a full workgroup
    if (!min_fill || !resident || n_min <= resident) return full;
    const uint64_t rounds = (n_min + resident - 1) / resident;
    const uint64_t want = (units + rounds * resident - 1) / (rounds * resident);
    return want * 100 >= (uint64_t)full * (uint64_t)min_fill ? (uint32_t)want : full;
}
static std::atomic<uint32_t> g_force_per_wg{0}, g_force_tiles{0};     // (svt_debug_force_wg: measurements)
extern "C" void svt_debug_force_wg(uint32_t per_wg, uint32_t tiles) { g_force_per_wg = per_wg; g_force_tiles = tiles; }
// Launches of less than one round: five-wave workgroups whose producers look up and whose consumer sums (svt_coop_kernel.h).
// A workgroup takes 64 ... 256 units -- as few as keep the launch inside ONE round of the resident cooperative workgroups, so
// that a launch of a few thousand units still spreads over the chip.
#ifndef SVT_COOP_MAX_UNITS
#define SVT_COOP_MAX_UNITS (1ull << 40)   /* (svt_debug_coop: measurements; the rule is SVT_COOP_CU_UNITS per CU) */
#endif
static std::atomic<uint64_t> g_coop_max_units{std::getenv("SVT_NO_COOP") ? uint64_t(0) : uint64_t(SVT_COOP_MAX_UNITS)};
static std::atomic<uint32_t> g_coop_per_wg{0};
extern "C" void svt_debug_coop(uint64_t max_units, uint32_t per_wg) { g_coop_max_units = max_units; g_coop_per_wg = per_wg; }   // (measurements)
// Which kernel a launch of less than one round takes (measurements: SVT_SMALL_KIND at build time, svt_debug_small_kind at run time):
// 0 = the rule below, 1 = the streaming kernel always, 2 = cooperative, 3 / 4 = two / four lanes per unit.
#ifndef SVT_SMALL_KIND
#define SVT_SMALL_KIND 0
#endif
// The rule (tools/small_kinds.py over 2 k ... 160 k units, profiles/r05_small_kinds.txt; 256 CUs, 100 records per unit, ms):
//   units    stream   coop    2 lanes  4 lanes
//   10 000   0.0358   0.0206  0.0364   0.0253      <= one cooperative workgroup of 64 units per CU: cooperative
//   30 000   0.0372   0.0296  0.0377   0.0261      <= one 4-lane workgroup of 256 units per CU: four lanes per unit
//   65 000   0.0387   0.0424  0.0390   0.0273
//   90 000   0.0490   0.0528  0.0469   0.0493      <= two 2-lane workgroups per CU: two lanes per unit (classic association;
//  131 000   0.0550   0.0639  0.0528   0.0536         the singlesample one spills at 128 registers: streaming kernel)
//  160 000   0.0684   0.0954  0.0826   0.0740      beyond: the streaming kernel
// Units of 400 records: 0.118 / 0.057 / 0.124 / 0.083 at 10 000 units -- the longer the units, the more the shorter chain is worth.
#ifndef SVT_SPLIT4_CU_UNITS
#define SVT_SPLIT4_CU_UNITS 256     // units per CU up to which a launch takes four lanes per unit (0 = never)
#endif
#ifndef SVT_SPLIT2_CU_UNITS
#define SVT_SPLIT2_CU_UNITS 512     // ... two lanes per unit
#endif
#ifndef SVT_COOP_CU_UNITS
#define SVT_COOP_CU_UNITS 64        // ... the cooperative kernel
#endif
static std::atomic<int> g_small_kind{SVT_SMALL_KIND};
extern "C" int svt_debug_small_kind(int kind) { return g_small_kind.exchange(kind); }
// lanes per unit for a launch over `units` units (0 = not the split kernel)
int split_lanes_for(const svt_batch* b, uint64_t units)
{
    const int kind = g_small_kind.load(std::memory_order_relaxed);
    const bool sso = (b->flags & SVT_FLAG_SSO_ASSOCIATION) != 0;
    if (!b->split_lds_bytes || !units) return 0;
    if (kind == 3) return sso && b->mode == kMultiLds ? 0 : 2;
    if (kind == 4) return 4;
    if (kind != 0) return 0;
    const uint64_t cus = cu_count(b->device);
    const bool coop_first = b->coop_lds_bytes && units <= std::min<uint64_t>(cus * SVT_COOP_CU_UNITS, g_coop_max_units.load(std::memory_order_relaxed));
    if (coop_first) return 0;
    return units <= cus * SVT_SPLIT4_CU_UNITS ? 4 : units <= cus * SVT_SPLIT2_CU_UNITS && !sso ? 2 : 0;
}

WgPlan wg_plan(const svt_batch* b, uint64_t units)
{
    WgPlan p;
    p.coop = false;
    p.split = 0;
    {
        const int kind = g_small_kind.load(std::memory_order_relaxed);
        const int lanes = split_lanes_for(b, units);
        if (lanes && b->split_lds_bytes && units && !g_force_per_wg.load(std::memory_order_relaxed)) {
            p.split = lanes;
            p.tiles = 1;
            p.per_wg = (uint32_t)kBlock;
            p.n_wg = (uint32_t)((units + p.per_wg - 1) / p.per_wg);
            return p;
        }
        if (kind == 1 || kind == 3 || kind == 4) goto stream;
    }
    if (b->coop_lds_bytes && units && g_small_kind.load(std::memory_order_relaxed) != 1 &&
        (units <= std::min<uint64_t>((uint64_t)cu_count(b->device) * SVT_COOP_CU_UNITS, g_coop_max_units.load(std::memory_order_relaxed)) ||
         g_small_kind.load(std::memory_order_relaxed) == 2) &&
        !g_force_per_wg.load(std::memory_order_relaxed)) {
        p.coop = true;
        p.per_wg = (uint32_t)kBlock;
        if (const uint32_t f = g_coop_per_wg.load(std::memory_order_relaxed)) p.per_wg = std::min<uint32_t>((f + 63u) / 64u * 64u, (uint32_t)kBlock);
        else
            for (uint32_t per = 64; per < (uint32_t)kBlock; per += 64)
                if ((units + per - 1) / per <= std::max<uint32_t>(b->coop_resident, 1)) { p.per_wg = per; break; }
        p.tiles = (int)(p.per_wg / 64u);     // (slots_of_launch: n_wg * tiles * 64 result slots)
        p.n_wg = (uint32_t)((units + p.per_wg - 1) / p.per_wg);
        return p;
    }
stream:
    if (const uint32_t f = g_force_per_wg.load(std::memory_order_relaxed)) {
        const int ft = (int)g_force_tiles.load(std::memory_order_relaxed);
        p.tiles = b->mode == kSingleLds && (ft == 1 || ft == 2) ? ft : tiles_per_wave(b, units);
        p.per_wg = std::min<uint32_t>(f, (uint32_t)kBlock * (uint32_t)p.tiles);
        p.n_wg = (uint32_t)((units + p.per_wg - 1) / p.per_wg);
        return p;
    }
    p.tiles = tiles_per_wave(b, units);
    const uint32_t full = (uint32_t)kBlock * (uint32_t)p.tiles;
    p.per_wg = balanced_units_per_wg(units, (units + full - 1) / full, full, b->resident_wgs);
    p.n_wg = (uint32_t)((units + p.per_wg - 1) / p.per_wg);
    return p;
}

uint64_t slots_of_launch(const svt_batch* b, uint64_t units)
{
    if (units == 0) return 0;
    if (b->layout == kLayoutPacked) return (units + kBlock - 1) / kBlock * kBlock;
    const WgPlan p = wg_plan(b, units);
    if (p.coop) return (uint64_t)p.n_wg * (uint64_t)p.tiles * (uint64_t)kWave;
    return (uint64_t)p.n_wg * (uint64_t)kBlock * (uint64_t)p.tiles;
}

int launch_stream(svt_batch* b, StreamArgs& a, hipStream_t stream)
{
    const uint64_t units = (uint64_t)a.unit_end - a.unit_begin;
    const WgPlan p = wg_plan(b, units);
    a.units_per_wg = p.per_wg;
    if (p.split) {
        StreamArgs c = a;
        c.lds_rings = b->split_region;
        c.l10_where = b->split_l10_where;
        c.lds_l10 = b->split_lds_l10;
        c.l10_lds_entries = b->split_l10_entries;
        const dim3 grid(p.n_wg), block(kBlock * p.split);
        void* params[] = {&c};
        HIP_TRY(hipLaunchKernel(split_kernel_of(b, p.split), grid, block, params, b->split_lds_bytes, stream));
        return SVT_OK;
    }
    if (p.coop) {
        StreamArgs c = a;
        c.lds_rings = b->coop_region;
        c.l10_where = b->coop_l10_where;
        c.lds_l10 = b->coop_lds_l10;
        c.l10_lds_entries = b->coop_l10_entries;
        const dim3 grid(p.n_wg), block(kCoopBlock);
        void* params[] = {&c};
        HIP_TRY(hipLaunchKernel(coop_kernel_of(b), grid, block, params, b->coop_lds_bytes, stream));
        return SVT_OK;
    }
    const dim3 grid(p.n_wg), block(kBlock);
    void* params[] = {&a};
    HIP_TRY(hipLaunchKernel(stream_kernel_of(b, p.tiles), grid, block, params, b->lds_bytes, stream));
    return SVT_OK;
}

// the pass over packed evidence: one library (tables in LDS) / several (library switches, tables through L2)
const void* packed_kernel_of(const svt_batch* b)
{
    const bool sso = (b->flags & SVT_FLAG_SSO_ASSOCIATION) != 0, multi = b->pargs.n_libs > 1;
    return sso ? (multi ? reinterpret_cast<const void*>(&svt_packed_kernel<true, 1, true>) : reinterpret_cast<const void*>(&svt_packed_kernel<true, 1, false>))
               : (multi ? reinterpret_cast<const void*>(&svt_packed_kernel<false, 1, true>) : reinterpret_cast<const void*>(&svt_packed_kernel<false, 1, false>));
}

// units [u0, u1) of a streamed layout (stream: not the library-window mode, whose launch covers window chunks);
// slot_begin: where this launch's tagged result records start (SVT_FLAG_RESULT96; slots_of_launch(b, u1 - u0) of them)
int launch_range(svt_batch* b, uint64_t u0, uint64_t u1, hipStream_t stream, uint64_t slot_begin = 0)
{
    if (u1 <= u0) return SVT_OK;
    if (b->layout == kLayoutPacked) {
        PackedArgs a = b->pargs;
        a.unit_begin = (uint32_t)u0;
        a.unit_end = (uint32_t)u1;
        a.slot_begin = (uint32_t)slot_begin;
        const dim3 grid((unsigned)((u1 - u0 + kBlock - 1) / kBlock)), block(kBlock);
        void* params[] = {&a};
        HIP_TRY(hipLaunchKernel(packed_kernel_of(b), grid, block, params, b->lds_bytes, stream));
        return SVT_OK;
    }
    StreamArgs a = b->sargs;
    a.unit_begin = (uint32_t)u0;
    a.unit_end = (uint32_t)u1;
    a.slot_begin = (uint32_t)slot_begin;
    return launch_stream(b, a, stream);
}

int ensure_result_slots(svt_batch* b, uint64_t slots);

int launch_genotype(svt_batch* b)
{
    if (b->layout == kLayoutPacked) {
        if (b->n_units == 0) return SVT_OK;
        const dim3 grid((unsigned)((b->n_units + kBlock - 1) / kBlock)), block(kBlock);
        void* params[] = {&b->pargs};
        HIP_TRY(hipLaunchKernel(packed_kernel_of(b), grid, block, params, b->lds_bytes, b->stream));
        return SVT_OK;
    }
    if (b->n_units == 0) return SVT_OK;
    if (b->mode == kMultiLds && b->window_tiles == 1) {
        // a launch of less than one round: K lanes per unit (svt_split_kernel.h; the chunks hold at most 256 units)
        if (const int lanes = split_lanes_for(b, b->n_units)) {
            StreamArgs c = b->sargs;
            c.lds_rings = b->split_region;
            c.l10_where = b->split_l10_where;
            c.lds_l10 = b->split_lds_l10;
            c.l10_lds_entries = b->split_l10_entries;
            c.chunk_begin = 0;
            const dim3 grid(b->n_chunks), block(kBlock * lanes);
            void* params[] = {&c};
            HIP_TRY(hipLaunchKernel(split_kernel_of(b, lanes), grid, block, params, b->split_lds_bytes, b->stream));
            return SVT_OK;
        }
    }
    if (b->mode == kMultiLds && b->split_window_kinds) {
        // two launches, one per kind of window: each kernel holds ONE record consumer (126 VGPRs: four workgroups per CU; the
        // kernel with both consumers has 161: three).  The chunks are ordered by the size of their window.
        const dim3 block(kBlock);
        if (b->n_chunks_one) {
            StreamArgs a = b->sargs;
            a.chunk_begin = 0;
            void* params[] = {&a};
            HIP_TRY(hipLaunchKernel(window_kernel_of_kind(1), dim3(b->n_chunks_one), block, params, b->lds_bytes, b->stream));
        }
        if (b->n_chunks > b->n_chunks_one) {
            StreamArgs a = b->sargs;
            a.chunk_begin = b->n_chunks_one;
            void* params[] = {&a};
            HIP_TRY(hipLaunchKernel(window_kernel_of_kind(2), dim3(b->n_chunks - b->n_chunks_one), block, params, b->lds_bytes, b->stream));
        }
        return SVT_OK;
    }
    if (b->mode != kMultiLds) {
        // The workgroup plan is looked up per launch (the debug hooks can move it between svt_batch_create and a pass): the tagged
        // records of THIS launch must fit what the result buffer was sized for -- the library's own buffer grows, a caller's does not.
        if (b->sargs.result96) {
            const uint64_t need = slots_of_launch(b, b->n_units);
            if (need != b->out_slots) {
                if (b->out_dev != b->d_out) {
                    if (need > b->bound_slots) return fail(SVT_ERR_STATE, "the pass needs more result slots than the bound device buffer holds");
                } else {
                    SVT_TRY(ensure_result_slots(b, need));
                }
                b->out_slots = need;
            }
        }
        return launch_stream(b, b->sargs, b->stream);
    }
    const dim3 grid(b->n_chunks), block(kBlock);   // library windows: one workgroup per chunk of a window's units
    void* params[] = {&b->sargs};
    HIP_TRY(hipLaunchKernel(stream_kernel_of(b, b->window_tiles), grid, block, params, b->lds_bytes, b->stream));
    return SVT_OK;
}

// SVT_TRACE=1 in the environment prints the stage times of svt_batch_create to stderr
struct StageTimer {
    bool on = std::getenv("SVT_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[svt] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// the record-contract violations svt_stream_kernel reports
int record_error(uint32_t err_bits)
{
    return fail(SVT_ERR_INVALID, record_error_text(err_bits));
}

// kLayoutStream: has the last pass seen a record that breaks the contract?  (blocking)
int check_stream_errors(svt_batch* b)
{
    if (b->layout != kLayoutStream || !b->d_err) return SVT_OK;
    uint32_t bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, b->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return bits ? record_error(bits) : SVT_OK;
}

// the result buffer must hold `slots` device records of this batch's form (SVT_FLAG_RESULT96: whole workgroups of 96-byte
// records, which for many small window chunks or many small launches can be more than n_units * 128 bytes)
int ensure_result_slots(svt_batch* b, uint64_t slots)
{
    const uint64_t bytes = std::max<uint64_t>(slots, 1) * ((b->flags & SVT_FLAG_RESULT96) ? sizeof(svt_result96) : sizeof(svt_result));
    if (bytes <= b->cap_out) return SVT_OK;
    HIP_TRY(hipStreamSynchronize(b->stream));
    const bool bound = b->out_dev != b->d_out;
    g_pool.put(b->device, b->d_out, b->cap_out);
    b->d_out = nullptr;
    b->cap_out = 0;
    void* p = nullptr;
    SVT_TRY(g_pool.get(b->device, bytes, &p, &b->cap_out));
    b->d_out = static_cast<svt_result*>(p);
    if (!bound) {
        b->out_dev = b->d_out;
        b->sargs.out = b->d_out;
        b->pargs.out = b->d_out;
    }
    return SVT_OK;
}

// svt_batch_create for the streaming layout: validate the unit arrays, build the tables, put the canonical
// CSR in HBM as it is.  No scan, no tiling, no re-encoding: the pass reads the records where they lie.
// `d_records_resident` (from the geometry stage) is adopted: the batch then owns that pool buffer.
// records (and units) one resident batch may hold: the kernels index both with 32 bits.  SVT_MAX_BATCH_RECORDS lowers it (tests
// of the chunked one-shot at sizes a test can afford).
uint64_t max_batch_records()
{
    static const uint64_t cached = [] {
        uint64_t v = 0xFFFFFFF0ull - 1;
        if (const char* e = std::getenv("SVT_MAX_BATCH_RECORDS")) {
            const uint64_t w = std::strtoull(e, nullptr, 10);
            if (w > 0 && w < v) v = w;
        }
        return v;
    }();
    return cached;
}

int create_stream(const svt_evidence_batch* in, svt_batch* b, void* d_records_resident = nullptr, uint64_t resident_cap = 0,
                  bool defer_records = false)   // defer_records: the caller uploads the records itself (pipelined one-shot)
{
    const uint64_t n = in->n_units;
    const uint64_t n_rec = n ? in->rec_offset[n] : 0;
    StageTimer tm;
    if (n_rec > max_batch_records())
        return fail(SVT_ERR_INVALID, "too many records in one batch (< 2^32): cut it with svt_chunk_bounds, or hand it to svt_genotype, which does");
    uint64_t max_f = 0;
    bool wide_var_length = false, all_hinted = n > 0;
    {   // the unit arrays, checked by several host threads
        const uint64_t kChunk = 16384, n_chunks = (n + kChunk - 1) / kChunk;
        struct Part { uint64_t max_f = 0; int bad = 0; bool wide = false, hinted = true; };
        std::vector<Part> parts(std::max<uint64_t>(n_chunks, 1));
        parallel_for(n_chunks, [&](uint64_t ch) {
            Part p;
            for (uint64_t u = ch * kChunk; u < std::min(n, (ch + 1) * kChunk); ++u) {
                if (in->rec_offset[u + 1] < in->rec_offset[u]) { p.bad |= 1; continue; }
                const uint64_t f = in->rec_offset[u + 1] - in->rec_offset[u];
                if (f > 0x3FFFFFFFull) p.bad |= 2;
                const svt_unit& U = in->units[u];
                if (U.svtype > SVT_SVTYPE_BND) p.bad |= 4;
                if ((U.libs >> 16) != 0 || (U.flags & ~SVT_UNIT_SKIP)) p.bad |= 8;
                const uint32_t w_lo = U.libs & 0xffu, w_cnt = (U.libs >> 8) & 0xffu;
                if (w_cnt && w_lo + w_cnt > in->n_libs) p.bad |= 16;
                p.hinted = p.hinted && w_cnt != 0;
                if (U.var_length < -(1 << 30) || U.var_length > (1 << 30)) p.wide = true;
                p.max_f = std::max(p.max_f, f);
            }
            parts[ch] = p;
        });
        int bad = 0;
        for (uint64_t ch = 0; ch < n_chunks; ++ch) {
            bad |= parts[ch].bad;
            max_f = std::max(max_f, parts[ch].max_f);
            wide_var_length = wide_var_length || parts[ch].wide;
            all_hinted = all_hinted && parts[ch].hinted;
        }
        if (bad & 1) return fail(SVT_ERR_INVALID, "rec_offset not monotone");
        if (bad & 2) return fail(SVT_ERR_INVALID, "unit with too many records");
        if (bad & 4) return fail(SVT_ERR_INVALID, "bad svtype");
        if (bad & 8) return fail(SVT_ERR_INVALID, "unit reserved/flags bits must be 0");
        if (bad & 16) return fail(SVT_ERR_INVALID, "unit library window beyond n_libs");
    }
    tm.mark("validate units");
    HostTables T;
    SVT_TRY(build_tables(in, max_f, T));
    if (wide_var_length) T.fast_geometry = false;

    // ---- several libraries: when every unit says which libraries its sample owns (svt_unit.libs), group the
    // units by that window -- a permutation of 4 bytes per unit, the records stay where they are -- and cut the
    // groups into workgroup chunks; a workgroup then stages only its window's histograms
    // library windows: two tiles per wave for launches that need more than one round of resident workgroups anyway
    // (the same rule and the same reason as tiles_per_wave for one library)
    b->window_tiles = (SVT_STREAM_R == 1 && SVT_WINDOW_TILES == 2 && n >= kTwoTilesMinUnits) ? 2 : SVT_STREAM_R;
    // classic association, two tiles per wave: the pass over library windows as two launches, a kernel per kind of window (one
    // record consumer each: 126 VGPRs, four workgroups per CU, against 161 / three for the kernel that holds both).  Measured on
    // the configs[4] batch at 2 M units, same memory (profiles/r05_window_split_ab.txt): 0.713 against 0.655 ms with one to three
    // libraries per sample -- two launches one after the other pay two ramp-downs --, 0.6144 against 0.6147 when every sample has
    // one library (one launch either way: the fourth workgroup per CU buys nothing here).  Off.
#ifndef SVT_WINDOW_SPLIT
#define SVT_WINDOW_SPLIT 0
#endif
    const bool split_kinds = SVT_WINDOW_SPLIT && b->window_tiles == 2 && !(b->flags & SVT_FLAG_SSO_ASSOCIATION) && !std::getenv("SVT_NO_WINDOW_SPLIT");
    auto window_budget_kernel = [&]() { return split_kinds ? window_kernel_of_kind(2) : stream_kernel_of(b, b->window_tiles); };
    const uint32_t kUnitsPerWg = (uint32_t)kBlock * (uint32_t)b->window_tiles;
    std::vector<uint32_t> perm;
    std::vector<uint2> chunks;
    std::vector<WgDesc> windows;
    struct Group { uint32_t begin, end; WgDesc w; };   // positions [begin, end) of perm: the units of one library window
    std::vector<Group> groups;
    uint32_t max_win_bins = 0, max_win_libs = 0;
    // Without hints (on every unit) the only window that is known to hold every record's library is the whole batch:
    // a run with a handful of libraries (one sample with 2-3 read-group libraries) still fits LDS that way; a joint
    // batch of many samples does not and needs the hints (else: general mode, tables through L2).
    uint64_t all_bins = 0;
    for (const LibDesc& L : T.libs) all_bins += L.n_bins + 1;
    const bool whole_batch_window = !all_hinted && n > 0 && in->n_libs <= 255 &&
                                    kSBins + all_bins * 4 + in->n_libs * sizeof(WinLib) + 64 + kWavesPerBlock * kStreamRingBytes <= (160 * 1024 / 2);
    const uint32_t whole_key = in->n_libs << 8;   // SVT_UNIT_LIBS(0, n_libs)
    const bool may_window = in->n_libs > 1 && T.fast_geometry && !(b->flags & SVT_FLAG_GENERAL_TABLES);
    bool windowed = may_window && (all_hinted || whole_batch_window);
    // No hints and too many libraries for one window: the windows are read off the records themselves, on the device,
    // right after the upload (svt_window_scan_kernel.h) -- not when the caller uploads the records later (pipelined one-shot)
    const bool derive_windows = may_window && !windowed && n > 0 && in->n_libs <= 255 && T.narrow_bins;
    // (the pipelined one-shot of such a batch: its pass is a few tenths of a millisecond beside tens of milliseconds of upload, so
    // nothing is lost by uploading first and reading the windows -- the general mode it used to take instead runs at a third of
    // the window kernel's speed)
    if (derive_windows) defer_records = false;
    b->records_resident = !defer_records;
    // group the units by window key (a counting sort: stable, original order inside a group) and cut the groups into chunks
    auto group_units = [&](auto&& key_of) {
        std::vector<uint32_t> start(65537, 0u);
        for (uint64_t u = 0; u < n; ++u) ++start[key_of(u) + 1];
        for (uint32_t k = 0; k < 65536u; ++k) start[k + 1] += start[k];
        perm.resize(n);
        groups.clear();
        {
            std::vector<uint32_t> at(start.begin(), start.end() - 1);
            for (uint64_t u = 0; u < n; ++u) perm[at[key_of(u)]++] = (uint32_t)u;
        }
        for (uint32_t k = 0; k < 65536u; ++k) {
            if (start[k + 1] == start[k]) continue;
            const uint32_t lo = k & 0xffu, cnt = k >> 8;
            WgDesc w{};
            w.lib_lo = lo;
            w.lib_cnt = cnt;
            w.bin_lo = T.libs[lo].tab_off;
            w.bin_cnt = T.libs[lo + cnt - 1].tab_off + T.libs[lo + cnt - 1].n_bins + 1 - w.bin_lo;
            max_win_bins = std::max(max_win_bins, w.bin_cnt);
            max_win_libs = std::max(max_win_libs, w.lib_cnt);
            groups.push_back(Group{start[k], start[k + 1], w});
        }
    };
    // ... and the groups into workgroup chunks of at most `per_chunk` units, the chunks of a group of (nearly) equal size
    auto cut_chunks = [&](const uint32_t per_chunk) {
        chunks.clear();
        windows.clear();
        for (const Group& g : groups) {
            const uint32_t units = g.end - g.begin, pieces = (units + per_chunk - 1) / per_chunk;
            for (uint32_t i = 0; i < pieces; ++i) {
                const uint32_t p0 = g.begin + (uint32_t)((uint64_t)units * i / pieces), p1 = g.begin + (uint32_t)((uint64_t)units * (i + 1) / pieces);
                chunks.push_back(make_uint2(p0, p1 - p0));
                windows.push_back(g.w);
            }
        }
    };
    if (windowed) {
        group_units([&](uint64_t u) -> uint32_t { return all_hinted ? in->units[u].libs & 0xffffu : whole_key; });
        tm.mark("group units by library window");
    }
    const uint32_t n_l10 = (uint32_t)T.l10.size();
    T.l10.resize(((size_t)n_l10 + 127) / 128 * 128, 0.0);   // the ring copy of the table moves whole KiB
    tm.mark("build tables");

    SVT_TRY(g_handles.get_stream(&b->stream));
    SVT_TRY(g_handles.get_event(&b->ev0, true));
    SVT_TRY(g_handles.get_event(&b->ev1, true));

    const uint64_t n_blk = std::max<uint64_t>((n_rec + kBlockRecords - 1) / kBlockRecords, 1);
    {
        Stager st(b->stream);
        void* p = nullptr;
        if (d_records_resident) {
            if (resident_cap < n_blk * 128) return fail(SVT_ERR_INTERNAL, "resident record buffer too small");
            b->d_records = d_records_resident;
            b->cap_records = resident_cap;
        } else {
            SVT_TRY(g_pool.get(b->device, n_blk * 128, &p, &b->cap_records, /*records=*/true));
            b->d_records = p;
        }
        // the tail of the last 128-byte block is read (and contract-checked) like any record: zero it
        if (n_blk * 128 > n_rec * 16)
            HIP_TRY(hipMemsetAsync(static_cast<char*>(b->d_records) + n_rec * 16, 0, n_blk * 128 - n_rec * 16, b->stream));
        if (!d_records_resident && !defer_records) SVT_TRY(st.copy(b->d_records, in->records, n_rec * sizeof(uint4)));
        SVT_TRY(g_pool.get(b->device, (n + 1) * sizeof(uint64_t), &p, &b->cap_off));
        b->d_off = static_cast<uint64_t*>(p);
        if (n) SVT_TRY(st.copy(b->d_off, in->rec_offset, (n + 1) * sizeof(uint64_t)));
        SVT_TRY(g_pool.get(b->device, std::max<uint64_t>(n, 1) * sizeof(svt_unit), &p, &b->cap_units));
        b->d_units = static_cast<svt_unit*>(p);
        SVT_TRY(st.copy(b->d_units, in->units, n * sizeof(svt_unit)));
        SVT_TRY(upload(&b->d_libs, T.libs, st));
        SVT_TRY(upload(&b->d_pm, T.pm, st));
        SVT_TRY(upload(&b->d_l10, T.l10, st));
        SVT_TRY(upload(&b->d_bins, T.bins, st));
        SVT_TRY(upload(&b->d_wtab, T.wtab, st));
        SVT_TRY(g_pool.get(b->device, std::max<uint64_t>(n, 1) * sizeof(svt_result), &p, &b->cap_out));
        b->d_out = static_cast<svt_result*>(p);
        SVT_TRY(g_handles.get_small(sizeof(uint32_t), &p));
        b->d_err = static_cast<uint32_t*>(p);
        HIP_TRY(hipMemsetAsync(b->d_err, 0, sizeof(uint32_t), b->stream));
        SVT_TRY(st.finish());
    }
    tm.mark("H2D CSR + tables (staged)");

    if (derive_windows) {
        // the scan's output borrows the buffer of the permutation it leads to
        void* pp = nullptr;
        SVT_TRY(g_pool.get(b->device, n * sizeof(uint32_t), &pp, &b->cap_perm));
        b->d_perm = static_cast<uint32_t*>(pp);
        const uint32_t n32 = (uint32_t)n;
        const unsigned waves = (unsigned)std::min<uint64_t>(n, 256ull * 32);          // the waves one pass of the chip holds
        const dim3 grid((waves + kScanBlock / kWave - 1) / (kScanBlock / kWave)), block(kScanBlock);
        hipLaunchKernelGGL(svt_window_scan_kernel, grid, block, 0, b->stream, static_cast<const uint4*>(b->d_records), b->d_off, n32, b->d_perm);
        HIP_TRY(hipGetLastError());
        std::vector<uint32_t> seen(n);
        HIP_TRY(hipMemcpyAsync(seen.data(), b->d_perm, n * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        tm.mark("library windows from the records (device scan)");
        // (0: a unit whose libraries lie > 255 apart; a window beyond n_libs: some record names a library the batch does
        // not have -- the general mode reports it as the contract violation it is)
        bool all_seen = true;
        for (uint64_t u = 0; u < n && all_seen; ++u) all_seen = seen[u] != 0u && (seen[u] & 0xffu) + ((seen[u] >> 8) & 0xffu) <= in->n_libs;
        if (all_seen) {
            group_units([&](uint64_t u) -> uint32_t { return seen[u] & 0xffffu; });
            windowed = true;
            tm.mark("group units by library window");
        }
    }

    // one library whose tables fit beside the rings: tables in LDS, 32-bit index math; anything else reads
    // the tables through L2 with exact 64-bit geometry
    size_t kStreamLdsPerWg = (160 * 1024 / 3) & ~size_t(127);   // three workgroups per CU (refined below: what the kernel's registers allow)
    constexpr size_t kStreamLdsPerWg2 = (160 * 1024 / 2) & ~size_t(127);  // two
    constexpr size_t kLdsBin = 2 * sizeof(uint16_t);   // thr + hist of one bin in LDS: 16-bit ranks
    const size_t single_lds = kSBins + T.bins.size() * kLdsBin;
    const bool single = in->n_libs == 1 && T.fast_geometry && T.narrow_bins && single_lds + kWavesPerBlock * kStreamRingBytes <= 96 * 1024 &&
                        !(b->flags & SVT_FLAG_GENERAL_TABLES);
    const size_t window_lds = kSBins + (((size_t)max_win_bins * kLdsBin + 15) & ~size_t(15)) + (size_t)max_win_libs * sizeof(WinLib);
    // (a window of more than 32 libraries: the kernel keeps one small-deletion gate bit per library of the window in a register)
    windowed = windowed && T.narrow_bins && max_win_libs <= 32 && window_lds + kWavesPerBlock * kStreamRingBytes <= kStreamLdsPerWg2;
    b->mode = single ? kSingleLds : windowed ? kMultiLds : kGeneral;
    // units that already come grouped by window (a sample-major batch, a one-window batch) need no permutation:
    // the kernel then walks the units themselves (no index loads in front of every unit header)
    bool identity = true;
    if (windowed) {
        // the chunks: whole rounds of equal workgroups (wg_plan's rule; what the window kernel's registers and this batch's
        // window tables + rings let a CU hold)
        uint32_t per_chunk = kUnitsPerWg;
        {
            int wgs = 3;
            hipFuncAttributes fa{};
            if (hipFuncGetAttributes(&fa, window_budget_kernel()) == hipSuccess && fa.numRegs > 0) wgs = std::max(1, std::min(8, 512 / ((fa.numRegs + 7) / 8 * 8)));
            else (void)hipGetLastError();
            const size_t lds = ((window_lds + 127) & ~size_t(127)) + kWavesPerBlock * kStreamRingBytes;
            const uint32_t resident = (uint32_t)std::min<size_t>((size_t)wgs, (160 * 1024) / lds) * cu_count(b->device);
            uint64_t n_min = 0;
            for (const Group& g : groups) n_min += (g.end - g.begin + kUnitsPerWg - 1) / kUnitsPerWg;
            per_chunk = balanced_units_per_wg(n, n_min, kUnitsPerWg, resident);
            // every group rounds its chunk count up: keep the total inside the rounds the rule aimed at
            if (per_chunk < kUnitsPerWg && resident) {
                const uint64_t rounds = (n_min + resident - 1) / resident;
                auto count = [&](uint32_t per) { uint64_t c = 0; for (const Group& g : groups) c += (g.end - g.begin + per - 1) / per; return c; };
                while (per_chunk < kUnitsPerWg && count(per_chunk) > rounds * resident) ++per_chunk;
            }
        }
        cut_chunks(per_chunk);
        Stager st(b->stream);
        void* pp = nullptr;
        for (uint64_t u = 0; u < n && identity; ++u) identity = perm[u] == (uint32_t)u;
        if (!identity) {
            if (!b->d_perm) {
                SVT_TRY(g_pool.get(b->device, std::max<uint64_t>(n, 1) * sizeof(uint32_t), &pp, &b->cap_perm));
                b->d_perm = static_cast<uint32_t*>(pp);
            }
            SVT_TRY(st.copy(b->d_perm, perm.data(), n * sizeof(uint32_t)));
        }
        SVT_TRY(upload(&b->d_chunks, chunks, st));
        SVT_TRY(upload(&b->d_windows, windows, st));
        SVT_TRY(st.finish());
        b->n_chunks = (uint32_t)chunks.size();
        // the chunks come ordered by window key = first library | libraries << 8: windows of one library first
        uint32_t n_one = 0;
        while (n_one < b->n_chunks && windows[n_one].lib_cnt == 1u) ++n_one;
        bool ordered = true;
        for (uint32_t i = n_one; i < b->n_chunks && ordered; ++i) ordered = windows[i].lib_cnt != 1u;
        b->n_chunks_one = n_one;
        b->split_window_kinds = split_kinds && ordered;
    }
    if (b->d_perm && (!windowed || identity)) {   // (the scan's buffer when no permutation came of it)
        g_pool.put(b->device, b->d_perm, b->cap_perm);
        b->d_perm = nullptr;
        b->cap_perm = 0;
    }
    StreamArgs& a = b->sargs;
    a.records = static_cast<const uint4*>(b->d_records);
    a.rec_offset = b->d_off;
    a.units = b->d_units;
    a.pm = b->d_pm;
    a.l10 = b->d_l10;
    a.libs = b->d_libs;
    a.bins = b->d_bins;
    a.wtab = b->d_wtab;
    a.n_l10 = n_l10;
    a.n_libs = in->n_libs;
    a.total_bins = (uint32_t)T.bins.size();
    a.last_blk = (uint32_t)(n_blk - 1);
    a.lds_bins = single ? a.total_bins : windowed ? max_win_bins : 0u;
    a.lds_libs = single || windowed ? 0u : in->n_libs;
    a.perm = b->d_perm;
    a.chunks = b->d_chunks;
    a.windows = b->d_windows;
    a.lds_winlibs = (uint32_t)(kSBins + (((size_t)a.lds_bins * kLdsBin + 15) & ~size_t(15)));   // (WinLib is read as 16-byte halves)
    size_t tables = a.lds_winlibs + (size_t)a.lds_libs * sizeof(LibDesc) + (windowed ? (size_t)max_win_libs * sizeof(WinLib) : 0);
    tables = (tables + 127) & ~size_t(127);
    if (single) {   // one round of the one-tile kernel: what its registers and (tables + rings, the log10 table at most beside them) allow
        int wgs = 3;
        hipFuncAttributes fa{};
        if (hipFuncGetAttributes(&fa, stream_kernel_of(b, 1)) == hipSuccess && fa.numRegs > 0) wgs = std::max(1, std::min(8, 512 / ((fa.numRegs + 7) / 8 * 8)));
        else (void)hipGetLastError();
        const size_t by_lds = (160 * 1024) / (tables + kWavesPerBlock * kStreamRingBytes);
        b->one_tile_round_units = (uint64_t)std::min<size_t>((size_t)wgs, std::max<size_t>(by_lds, 1)) * cu_count(b->device) * kBlock;
    }
    // How many workgroups of this batch's kernel a CU can hold is decided by its registers (512 per SIMD lane: <= 128 VGPRs
    // = four waves per SIMD = four 256-thread workgroups per CU); the LDS budget per workgroup follows from that, so that
    // the tables never cost a workgroup the registers would allow.
    {
        int wgs = 3;
        hipFuncAttributes fa{};
        if (hipFuncGetAttributes(&fa, b->mode == kMultiLds ? window_budget_kernel() : stream_kernel_of(b, tiles_per_wave(b, n))) == hipSuccess && fa.numRegs > 0)
            wgs = std::max(1, std::min(8, 512 / ((fa.numRegs + 7) / 8 * 8)));
        else
            (void)hipGetLastError();
        if (const char* e = std::getenv("SVT_STREAM_WGS_PER_CU")) wgs = std::max(1, std::atoi(e));   // (measurements)
#ifdef SVT_FORCE_WGS
        wgs = SVT_FORCE_WGS;
#endif
        kStreamLdsPerWg = (160 * 1024 / (size_t)wgs) & ~size_t(127);
        b->wgs_per_cu = wgs;
    }
    // the log10 table of the epilogue: beside the tables when it costs no workgroup -- `fit` = what registers AND the
    // tables + rings allow --, else its first ring-stageful of entries through the wave's ring before each epilogue (a unit
    // whose read count reaches beyond them takes the table through L2), else through L2
    const size_t l10_bytes = ((size_t)n_l10 * 8 + 127) & ~size_t(127);
    const size_t base_lds = tables + kWavesPerBlock * kStreamRingBytes;
    const size_t fit = std::max<size_t>(1, std::min<size_t>((size_t)b->wgs_per_cu, (160 * 1024) / base_lds));
    const size_t budget = ((160 * 1024) / fit) & ~size_t(127);
    (void)kStreamLdsPerWg;
    if (base_lds + l10_bytes <= budget) {
        a.l10_where = kL10Shared;
        a.lds_l10 = (uint32_t)tables;
        a.l10_lds_entries = n_l10;
        tables += l10_bytes;
    } else if (kStreamDepth == 1 && SVT_L10_THROUGH_RING) {
        a.l10_where = kL10Ring;
        a.l10_lds_entries = (uint32_t)std::min<uint64_t>((n_l10 + 127u) / 128u * 128u, kStreamRingBytes / 8);   // (whole KiB move)
    } else {
        a.l10_where = kL10Global;
        a.l10_lds_entries = 0;
    }
    a.lds_rings = (uint32_t)tables;
    // the cooperative kernel (one library): its own region behind the tables, two workgroups per CU
    if (b->mode == kSingleLds && a.l10_where != kL10Ring) {
        size_t ctab = (a.lds_winlibs + 127) & ~size_t(127);     // the tables without the log10 table
        b->coop_l10_where = kL10Global;
        b->coop_lds_l10 = 0;
        b->coop_l10_entries = 0;
        if (ctab + l10_bytes + kCoopRegionBytes <= (160 * 1024 / 2)) {
            b->coop_l10_where = kL10Shared;
            b->coop_lds_l10 = (uint32_t)ctab;
            b->coop_l10_entries = n_l10;
            ctab += l10_bytes;
        }
        if (ctab + kCoopRegionBytes <= 160 * 1024) {
            b->coop_region = (uint32_t)ctab;
            b->coop_lds_bytes = ctab + kCoopRegionBytes;
            int wgs = 2;
            hipFuncAttributes fa{};
            if (hipFuncGetAttributes(&fa, coop_kernel_of(b)) == hipSuccess && fa.numRegs > 0) wgs = std::max(1, std::min(4, 512 / ((fa.numRegs + 7) / 8 * 8) * 4 / kCoopWaves));
            else (void)hipGetLastError();
            b->coop_resident = (uint32_t)std::min<size_t>((size_t)wgs, (160 * 1024) / b->coop_lds_bytes) * cu_count(b->device);
            if (b->coop_lds_bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute(coop_kernel_of(b), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->coop_lds_bytes));
        }
        // ... and the kernel with K lanes per unit
        size_t stab = (a.lds_winlibs + 127) & ~size_t(127);
        if (stab + l10_bytes + kSplitRegionBytes <= (160 * 1024 / 2)) {
            b->split_l10_where = kL10Shared;
            b->split_lds_l10 = (uint32_t)stab;
            b->split_l10_entries = n_l10;
            stab += l10_bytes;
        }
        if (stab + kSplitRegionBytes <= 160 * 1024) {
            b->split_region = (uint32_t)stab;
            b->split_lds_bytes = stab + kSplitRegionBytes;
            if (b->split_lds_bytes > 64 * 1024)
                for (int lanes = 2; lanes <= 4; lanes += 2)
                    HIP_TRY(hipFuncSetAttribute(split_kernel_of(b, lanes), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->split_lds_bytes));
        }
    }
    if (b->mode == kMultiLds && a.l10_where != kL10Ring) {
        // the kernel with K lanes per unit over library windows: its region behind the window tables (and the log10 table where
        // the streaming kernel keeps it beside them)
        const size_t stab = (tables + 127) & ~size_t(127);
        if (stab + kSplitRegionBytes <= 160 * 1024) {
            b->split_region = (uint32_t)stab;
            b->split_lds_bytes = stab + kSplitRegionBytes;
            b->split_l10_where = a.l10_where;
            b->split_lds_l10 = a.lds_l10;
            b->split_l10_entries = a.l10_lds_entries;
            if (b->split_lds_bytes > 64 * 1024)
                for (int lanes = 2; lanes <= 4; lanes += 2)
                    if (lanes == 4 || !(b->flags & SVT_FLAG_SSO_ASSOCIATION))
                        HIP_TRY(hipFuncSetAttribute(split_kernel_of(b, lanes), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->split_lds_bytes));
        }
    }
    a.n_units = n;
    a.unit_begin = 0;
    a.unit_end = (uint32_t)n;
    a.units_per_wg = (uint32_t)kBlock * (uint32_t)tiles_per_wave(b, n);
    a.result96 = (b->flags & SVT_FLAG_RESULT96) ? 1u : 0u;
    a.slot_begin = 0;
    b->resident_wgs = (uint32_t)std::min<size_t>((size_t)b->wgs_per_cu, (160 * 1024) / (tables + kWavesPerBlock * kStreamRingBytes + SVT_PROBE_LDS_PAD)) *
                      cu_count(b->device);
    b->out_dev = b->d_out;
    b->out_slots = !a.result96 ? n : b->mode == kMultiLds ? (uint64_t)b->n_chunks * kBlock * (uint64_t)b->window_tiles : slots_of_launch(b, n);
    SVT_TRY(ensure_result_slots(b, b->out_slots));
    a.out = b->d_out;
    a.err = b->d_err;
    a.lib0 = T.libs[0];
    fill_gt_consts(a.c, in->split_weight, in->disc_weight);
    b->out_dev = b->d_out;   // svt_batch_device_results / svt_batch_bind_device_results / svt_batch_site_qual
    b->lds_bytes = tables + kWavesPerBlock * kStreamRingBytes + SVT_PROBE_LDS_PAD;
    if (b->lds_bytes > 160 * 1024) return fail(SVT_ERR_INVALID, "LDS budget exceeded");
    if (tm.on)
        std::fprintf(stderr, "[svt] kernel budget: %d workgroups/CU by registers, LDS %zu B/workgroup (%zu fit), log10 table (%u entries) %s (%u entries)\n",
                     b->wgs_per_cu, b->lds_bytes, (size_t)(160 * 1024) / std::max<size_t>(b->lds_bytes, 1), n_l10,
                     a.l10_where == kL10Shared ? "in LDS" : a.l10_where == kL10Ring ? "through the ring" : "through L2", a.l10_lds_entries);
    if (b->lds_bytes > 64 * 1024)
        for (int tiles = 1; tiles <= 2; ++tiles)
            if (b->mode != kGeneral || tiles == 1)
                HIP_TRY(hipFuncSetAttribute(stream_kernel_of(b, tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes));
    if (b->lds_bytes > 64 * 1024 && b->split_window_kinds)
        for (int kind = 1; kind <= 2; ++kind)
            HIP_TRY(hipFuncSetAttribute(window_kernel_of_kind(kind), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes));
    return SVT_OK;
}

// ------------------------------------------------------------------------------------------
// packed evidence (include/svtyper_hip.h: svt_packed_evidence)
// ------------------------------------------------------------------------------------------
// Page-locked host buffers for the slots of packed evidence: hipHostMalloc of hundreds of MB costs tens of ms, so
// svt_packed_free hands the buffer back here (svt_trim releases them).  Without a device plain memory is used.
struct PinnedPool {
    struct Item { void* p; uint64_t cap; bool pinned; };
    std::mutex lock;
    std::vector<Item> idle, live;
    void* get(uint64_t bytes)
    {
        bytes = std::max<uint64_t>(bytes, 4096);
        std::lock_guard<std::mutex> g(lock);
        size_t best = idle.size();
        for (size_t i = 0; i < idle.size(); ++i)
            if (idle[i].cap >= bytes && idle[i].cap <= 2 * bytes + (1u << 20) && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
        Item it{};
        if (best != idle.size()) {
            it = idle[best];
            idle.erase(idle.begin() + (long)best);
        } else {
            it.cap = bytes + bytes / 8;
            int ndev = 0;
            if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipHostMalloc(&it.p, it.cap, hipHostMallocDefault) == hipSuccess) it.pinned = true;
            else {
                (void)hipGetLastError();
                it.p = std::malloc(it.cap);
                it.pinned = false;
            }
            if (!it.p) return nullptr;
        }
        live.push_back(it);
        return it.p;
    }
    // is [p, p + bytes) inside a live page-locked buffer of this pool?
    bool is_pinned(const void* p, uint64_t bytes = 1)
    {
        std::lock_guard<std::mutex> g(lock);
        const char* q = static_cast<const char*>(p);
        for (const Item& it : live)
            if (it.pinned && q >= static_cast<const char*>(it.p) && q + bytes <= static_cast<const char*>(it.p) + it.cap) return true;
        return false;
    }
    void put(void* p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> g(lock);
        for (size_t i = 0; i < live.size(); ++i)
            if (live[i].p == p) {
                idle.push_back(live[i]);
                live.erase(live.begin() + (long)i);
                break;
            }
        while (idle.size() > 4) {   // keep the largest
            size_t smallest = 0;
            for (size_t i = 1; i < idle.size(); ++i)
                if (idle[i].cap < idle[smallest].cap) smallest = i;
            release(idle[smallest]);
            idle.erase(idle.begin() + (long)smallest);
        }
    }
    static void release(const Item& it)
    {
        if (it.pinned) (void)hipHostFree(it.p);
        else std::free(it.p);
    }
    void trim()
    {
        std::lock_guard<std::mutex> g(lock);
        for (const Item& it : idle) release(it);
        idle.clear();
    }
};
PinnedPool g_pinned;

struct PackedOwner {             // what svt_pack_evidence returns: the public struct first, the storage behind it
    svt_packed_evidence pub{};
    uint32_t* off = nullptr;     // the three arrays that cross PCIe live in page-locked memory (g_pinned)
    svt_unit* units = nullptr;
    void* slots = nullptr;
    std::vector<std::vector<uint32_t>> hists;    // the libraries, copied: the evidence outlives the caller's batch
    std::vector<svt_library> libs;
    ~PackedOwner()
    {
        g_pinned.put(off);
        g_pinned.put(units);
        g_pinned.put(slots);
    }
};

// svt_pack_evidence: the encoder itself is host-only code in svt_pack.cpp; here it gets the page-locked pool as allocator
int pack_evidence(const svt_evidence_batch* in, svt_packed_evidence** out)
{
    if (!in || !out) return fail(SVT_ERR_INVALID, "null argument");
    *out = nullptr;
    const PackAlloc pool{[](uint64_t bytes) { return g_pinned.get(bytes); }, [](void* p) { g_pinned.put(p); }};
    PackedArrays arr;
    if (const char* e = std::getenv("SVT_PACK_TEST_RANGES")) {
        // (tests: the ranged form of the encoder -- what svt_genotype_packed_from_records drives -- without a consumer; the
        // arrays must be the plain call's)
        PackSink sink;
        sink.range_units = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
        sink.slots_cap = (in->n_units ? in->rec_offset[in->n_units] : 0) + 3 * in->n_units + 64;
        static thread_local uint64_t last_u1;
        last_u1 = 0;
        sink.ready = [](void*, const PackedArrays* a, uint64_t u0, uint64_t u1, uint64_t s0, uint64_t s1) -> int {
            if (u0 != last_u1 || u1 < u0 || s1 < s0 || (u1 > u0 && (a->off[3 * u0] != s0 || a->off[3 * u1] != s1))) return fail(SVT_ERR_INTERNAL, "ranged encoder: ranges out of order");
            last_u1 = u1;
            return SVT_OK;
        };
        SVT_TRY(encode_packed(in, pool, &arr, &sink));
        if (last_u1 != in->n_units) { g_pinned.put(arr.off); g_pinned.put(arr.units); g_pinned.put(arr.slots); return fail(SVT_ERR_INTERNAL, "ranged encoder: units missing"); }
    } else
    SVT_TRY(encode_packed(in, pool, &arr));
    auto owner = std::make_unique<PackedOwner>();
    owner->off = arr.off;
    owner->units = arr.units;
    owner->slots = arr.slots;
    owner->hists.resize(in->n_libs);
    owner->libs.assign(in->libs, in->libs + in->n_libs);
    for (uint32_t l = 0; l < in->n_libs; ++l) {
        owner->hists[l].assign(in->libs[l].hist, in->libs[l].hist + in->libs[l].n_bins);
        owner->libs[l].hist = owner->hists[l].data();
    }
    svt_packed_evidence& P = owner->pub;
    P.n_units = in->n_units;
    P.n_slots = arr.n_slots;
    P.n_records = arr.n_records;
    P.slot_offset = owner->off;
    P.units = owner->units;
    P.slots = owner->slots;
    P.common_mapq = arr.common;
    P.n_libs = in->n_libs;
    P.libs = owner->libs.data();
    P.split_weight = in->split_weight;
    P.disc_weight = in->disc_weight;
    *out = &owner.release()->pub;
    return SVT_OK;
}

// svt_batch_create_packed: upload the slots as they are + tables
// defer_all (svt_genotype_packed_from_records: the encoder is still running): `in` carries the library, the weights, the unit
// count and in n_slots the CAPACITY to allocate; slot offsets, unit headers and slots arrive later, range by range;
// max_f_known = the most records any unit has
int create_packed(const svt_packed_evidence* in, svt_batch* b, bool defer_slots = false, bool defer_all = false, uint64_t max_f_known = 0)
{
    const uint64_t n = in->n_units;
    StageTimer tm;
    if (in->n_libs == 0 || in->n_libs > 256 || !in->libs) return fail(SVT_ERR_INVALID, "n_libs must be 1..256");
    if (!defer_all) {
    if (n && (!in->slot_offset || !in->units)) return fail(SVT_ERR_INVALID, "null unit arrays");
    if (n && in->slot_offset[0] != 0) return fail(SVT_ERR_INVALID, "slot_offset[0] must be 0");
    if (n && in->slot_offset[3 * n] != in->n_slots) return fail(SVT_ERR_INVALID, "slot_offset does not end at n_slots");
    if (in->n_slots && !in->slots) return fail(SVT_ERR_INVALID, "null slots");
    }
    if (in->common_mapq > 0xffffu) return fail(SVT_ERR_INVALID, "common_mapq is two bytes");
    if (!(in->split_weight >= 0.0) || !(in->disc_weight >= 0.0) || !std::isfinite(in->split_weight) || !std::isfinite(in->disc_weight))
        return fail(SVT_ERR_INVALID, "weights must be finite and >= 0");

    SVT_TRY(g_handles.get_stream(&b->stream));
    SVT_TRY(g_handles.get_event(&b->ev0, true));
    SVT_TRY(g_handles.get_event(&b->ev1, true));
    // ---- the slots leave first (page-locked by svt_pack_evidence: straight DMA); the unit arrays are checked
    // while they are on the wire
    void* p = nullptr;
    SVT_TRY(g_pool.get(b->device, std::max<uint64_t>(in->n_slots, 1) * 16, &p, &b->cap_records));
    b->d_records = p;
    SVT_TRY(g_pool.get(b->device, (3 * n + 1) * sizeof(uint32_t), &p, &b->cap_soff));
    b->d_soff = static_cast<uint32_t*>(p);
    SVT_TRY(g_pool.get(b->device, std::max<uint64_t>(n, 1) * sizeof(svt_unit), &p, &b->cap_units));
    b->d_units = static_cast<svt_unit*>(p);
    SVT_TRY(g_pool.get(b->device, std::max<uint64_t>(n, 1) * sizeof(svt_result), &p, &b->cap_out));
    b->d_out = static_cast<svt_result*>(p);
    const bool slots_pinned = !defer_all && in->n_slots && g_pinned.is_pinned(in->slots, in->n_slots * 16);
    const bool off_pinned = !defer_all && n && g_pinned.is_pinned(in->slot_offset, (3 * n + 1) * sizeof(uint32_t));
    const bool units_pinned = !defer_all && n && g_pinned.is_pinned(in->units, n * sizeof(svt_unit));
    if (slots_pinned && !defer_slots) HIP_TRY(hipMemcpyAsync(b->d_records, in->slots, in->n_slots * 16, hipMemcpyHostToDevice, b->stream));
    if (off_pinned) HIP_TRY(hipMemcpyAsync(b->d_soff, in->slot_offset, (3 * n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
    if (units_pinned) HIP_TRY(hipMemcpyAsync(b->d_units, in->units, n * sizeof(svt_unit), hipMemcpyHostToDevice, b->stream));
    tm.mark("allocations + DMA enqueued");

    uint64_t max_f = max_f_known;   // bound of the records behind a unit: 8 pair entries, 7 weight entries per slot
    if (!defer_all) {
        const uint64_t kChunk = 16384, n_chunks = (n + kChunk - 1) / kChunk;
        std::vector<uint64_t> chunk_max(std::max<uint64_t>(n_chunks, 1), 0);
        std::vector<int> chunk_bad(std::max<uint64_t>(n_chunks, 1), 0);
        parallel_for(n_chunks, [&](uint64_t ch) {
            uint64_t m = 0;
            int bad = 0;
            for (uint64_t u = ch * kChunk; u < std::min(n, (ch + 1) * kChunk); ++u) {
                const svt_unit& U = in->units[u];
                const uint32_t* o = in->slot_offset + 3 * u;
                if (o[1] < o[0] || o[2] < o[1] || o[3] < o[2]) bad |= 1;
                if (U.svtype > SVT_SVTYPE_BND) bad |= 2;
                if ((U.libs >> 16) != 0 || (U.flags & ~SVT_UNIT_SKIP)) bad |= 4;
                if (U.var_length < -(1 << 30) || U.var_length > (1 << 30)) bad |= 8;
                if (U.svtype == SVT_SVTYPE_DEL && U.var_length < 0) bad |= 16;
                m = std::max(m, std::max<uint64_t>((uint64_t)(o[1] - o[0]) * 8, std::max<uint64_t>((uint64_t)(o[2] - o[1]) * 7, (uint64_t)(o[3] - o[2]) * 7)));
            }
            chunk_max[ch] = m;
            chunk_bad[ch] = bad;
        });
        int bad = 0;
        for (uint64_t ch = 0; ch < n_chunks; ++ch) { max_f = std::max(max_f, chunk_max[ch]); bad |= chunk_bad[ch]; }
        if (bad & 1) return fail(SVT_ERR_INVALID, "slot_offset not monotone");
        if (bad & 2) return fail(SVT_ERR_INVALID, "bad svtype");
        if (bad & 4) return fail(SVT_ERR_INVALID, "unit reserved/flags bits must be 0");
        if (bad & 8) return fail(SVT_ERR_UNSUPPORTED, "var_length outside the packed format's range");
        if (bad & 16) return fail(SVT_ERR_UNSUPPORTED, "negative DEL length");
    }
    svt_evidence_batch shell{};   // what build_tables looks at
    shell.n_units = 0;
    shell.n_libs = in->n_libs;
    shell.libs = in->libs;
    shell.split_weight = in->split_weight;
    shell.disc_weight = in->disc_weight;
    HostTables T;
    SVT_TRY(build_tables(&shell, max_f, T));
    // the limits of the packed format (include/svtyper_hip.h), library side; the unit side was checked above
    for (const LibDesc& L : T.libs)
        if (L.n_bins > kMaxShortBins) return fail(SVT_ERR_UNSUPPORTED, "histogram too wide for the packed pair entries");
    if (!T.fast_geometry) return fail(SVT_ERR_UNSUPPORTED, "library geometry outside the packed format's range");
    tm.mark("validate + tables");
    {
        Stager st(b->stream);
        if (in->n_slots && !slots_pinned && !defer_slots) SVT_TRY(st.copy(b->d_records, in->slots, in->n_slots * 16));
        if (n && !off_pinned && !defer_all) SVT_TRY(st.copy(b->d_soff, in->slot_offset, (3 * n + 1) * sizeof(uint32_t)));
        if (n && !units_pinned && !defer_all) SVT_TRY(st.copy(b->d_units, in->units, n * sizeof(svt_unit)));
        SVT_TRY(upload(&b->d_pm, T.pm, st));
        SVT_TRY(upload(&b->d_l10, T.l10, st));
        SVT_TRY(upload(&b->d_bins, T.bins, st));
        SVT_TRY(upload(&b->d_libs, T.libs, st));
        SVT_TRY(upload(&b->d_wtab, T.wtab, st));
        SVT_TRY(st.finish());
    }
    tm.mark("H2D slots + unit arrays + tables");
    const bool multi = in->n_libs > 1;     // library switches in the pair streams: descriptors in LDS, tables through L2
    b->mode = multi ? kGeneral : kSingleLds;
    b->n_slots = in->n_slots;
    PackedArgs& a = b->pargs;
    a.slots = static_cast<const uint4*>(b->d_records);
    a.slot_offset = b->d_soff;
    a.units = b->d_units;
    a.pm = b->d_pm;
    a.l10 = b->d_l10;
    a.bins = b->d_bins;
    a.libs = b->d_libs;
    a.n_libs = in->n_libs;
    a.wtab = b->d_wtab;
    a.n_l10 = (uint32_t)T.l10.size();
    a.total_bins = (uint32_t)T.bins.size();
    a.common_mq = in->common_mapq;
    size_t tables = kLdsBins + (multi ? (size_t)in->n_libs * sizeof(LibDesc) : (size_t)a.total_bins * sizeof(Bin));
    tables = (tables + 127) & ~size_t(127);
    constexpr size_t kLdsPerWg = (160 * 1024 / 3) & ~size_t(127);   // three workgroups per CU
    const size_t l10_bytes = ((size_t)a.n_l10 * 8 + 127) & ~size_t(127);
    a.lds_l10 = (uint32_t)tables;
    if (tables + l10_bytes + kWavesPerBlock * kRingBytes <= kLdsPerWg) {
        a.l10_where = kL10Shared;
        tables += l10_bytes;
    } else {
        a.l10_where = kL10Global;
    }
    a.lds_rings = (uint32_t)tables;
    a.n_units = n;
    a.unit_begin = 0;
    a.unit_end = (uint32_t)n;
    a.result96 = (b->flags & SVT_FLAG_RESULT96) ? 1u : 0u;
    a.slot_begin = 0;
    b->out_dev = b->d_out;
    b->out_slots = a.result96 ? slots_of_launch(b, n) : n;
    SVT_TRY(ensure_result_slots(b, b->out_slots));
    a.out = b->d_out;
    a.lib0 = T.libs[0];
    fill_gt_consts(a.c, in->split_weight, in->disc_weight);
    b->out_dev = b->d_out;
    b->lds_bytes = tables + kWavesPerBlock * kRingBytes;
    if (b->lds_bytes > 160 * 1024) return fail(SVT_ERR_INVALID, "LDS budget exceeded");
    if (tm.on)
        std::fprintf(stderr, "[svt] kernel budget: %d workgroups/CU by registers, LDS %zu B/workgroup (%zu fit), log10 table %s\n", b->wgs_per_cu,
                     b->lds_bytes, (size_t)(160 * 1024) / std::max<size_t>(b->lds_bytes, 1),
                     a.l10_where == kL10Shared ? "in LDS" : a.l10_where == kL10Ring ? "through the ring" : "through L2");
    if (b->lds_bytes > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute(packed_kernel_of(b), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes));
    }
    return SVT_OK;
}

// ------------------------------------------------------------------------------------------
// one shot, pipelined: H2D || kernel || D2H (the reference's 2-pass batch pipeline, singlesample.py:710-762,
// re-cast for one GPU).  The payload -- records or packed slots -- goes up in pieces of whole units on the batch's
// stream; a piece's units are genotyped on a second stream as soon as it has landed (one launch per piece) and,
// when the caller's output array is page-locked (svt_pinned_alloc), their result records go down on a third
// stream while the next piece is still on the wire.  PCIe is full duplex, so the wall time is the upload plus the
// last piece's pass and download.
// ------------------------------------------------------------------------------------------
struct PipeStreams {
    hipStream_t compute = nullptr, down = nullptr;
    std::vector<hipEvent_t> events;
    ~PipeStreams()
    {
        if (compute) (void)hipStreamSynchronize(compute);
        if (down) (void)hipStreamSynchronize(down);
        for (hipEvent_t e : events) g_handles.put_event(e, false);
        g_handles.put_stream(compute);
        g_handles.put_stream(down);
    }
    int event(hipEvent_t* e)
    {
        SVT_TRY(g_handles.get_event(e, false));
        events.push_back(*e);
        return SVT_OK;
    }
};


// ---- SVT_FLAG_RESULT96: 96-byte device records -> the caller's svt_result[] -----------------------------------------
static_assert(sizeof(svt_result96) == 96 && sizeof(svt_result) == 128, "result record sizes");
static_assert(offsetof(svt_result96, qr) == offsetof(svt_result, counts) && offsetof(svt_result96, gt) == 84, "svt_result96 is a prefix of svt_result + gt");

inline uint32_t result_bytes(const svt_batch* b) { return (b->flags & SVT_FLAG_RESULT96) ? (uint32_t)sizeof(svt_result96) : (uint32_t)sizeof(svt_result); }

// one record: the counts that are not in the 96-byte form are the reference's truncations of sums of the tallies
// (classic.py:455-469; the additions in its order, -ffp-contract=off on the host as on the device), 0 for blank / skipped units
inline void expand96_one(const svt_result96& r, svt_result& o)
{
    std::memcpy(&o, &r, 84);                     // gl, sq, tallies, QR, QA, GQ (the tag is not part of svt_result)
    const double ref_seq = r.tallies[SVT_TAL_REF_SEQ], alt_seq = r.tallies[SVT_TAL_ALT_SEQ], alt_clip = r.tallies[SVT_TAL_ALT_CLIP],
                 ref_span = r.tallies[SVT_TAL_REF_SPAN], alt_span = r.tallies[SVT_TAL_ALT_SPAN];
    const bool counted = r.gt >= 0 || r.gt == SVT_GT_MISSING;   // (a blank or skipped unit leaves every count 0)
    o.counts[SVT_CNT_DP] = counted ? (int32_t)(ref_seq + alt_seq + alt_clip + ref_span + alt_span) : 0;
    o.counts[SVT_CNT_RO] = counted ? (int32_t)(ref_seq + ref_span) : 0;
    o.counts[SVT_CNT_AO] = counted ? (int32_t)(alt_seq + alt_clip + alt_span) : 0;
    o.counts[SVT_CNT_RS] = counted ? (int32_t)ref_seq : 0;
    o.counts[SVT_CNT_AS] = counted ? (int32_t)alt_seq : 0;
    o.counts[SVT_CNT_ASC] = counted ? (int32_t)alt_clip : 0;
    o.counts[SVT_CNT_RP] = counted ? (int32_t)ref_span : 0;
    o.counts[SVT_CNT_AP] = counted ? (int32_t)alt_span : 0;
    o.gt = r.gt;
    std::memset(o.pad, 0, sizeof(o.pad));
}

// Tagged 96-byte records (SVT_FLAG_RESULT96: the kernel's order, padding tagged SVT_NO_UNIT) -> out[tag] as svt_result
// records; `in` and `out` disjoint; split over the host threads.  Whether every unit is covered EXACTLY once is tracked per
// unit (one byte each, claimed with an atomic exchange before the record is written): a tag out of range, or a second record
// for a unit, is refused on the spot -- nothing is written for it, no two threads ever write one out[] element -- and a
// unit nobody claimed shows in the count.  (A count and a sum of the tags, the first form, let {1, 1, 2, 2} pass for {0, 1, 2, 3}.)
struct Placed {
    uint64_t n_units = 0;
    std::unique_ptr<std::atomic<unsigned char>[]> seen;
    std::atomic<uint64_t> count{0};
    std::atomic<bool> bad{false};
    explicit Placed(uint64_t n) : n_units(n), seen(n ? new std::atomic<unsigned char>[n]() : nullptr) {}
    // true: the caller may write out[u]
    bool claim(uint32_t u)
    {
        if (u >= n_units || seen[u].exchange(1, std::memory_order_relaxed)) { bad.store(true, std::memory_order_relaxed); return false; }
        return true;
    }
    bool covers(uint64_t n) const { return n == n_units && !bad.load() && count.load() == n_units; }
};

inline void expand96(const svt_result96* in, uint64_t n, svt_result* out, Placed& placed)
{
    const uint64_t kChunk = 8192;
    const uint64_t chunks = (n + kChunk - 1) / kChunk;
    auto run = [&](uint64_t c) {
        const uint64_t hi = std::min(n, (c + 1) * kChunk);
        uint64_t mine = 0;
        for (uint64_t i = c * kChunk; i < hi; ++i) {
            const uint32_t u = in[i].unit;
            if (u == SVT_NO_UNIT || !placed.claim(u)) continue;
            expand96_one(in[i], out[u]);
            ++mine;
        }
        placed.count.fetch_add(mine, std::memory_order_relaxed);
    };
    if (chunks <= 1) { if (chunks) run(0); }
    else parallel_for(chunks, run);
}

// the batch's device result records -> out[n_units] (svt_result), whichever form the device holds
int d2h_results(svt_batch* b, svt_result* out)
{
    const uint64_t n = b->n_units;
    if (!n) return SVT_OK;
    if (!(b->flags & SVT_FLAG_RESULT96)) {
        if (g_pinned.is_pinned(out, n * sizeof(svt_result))) {   // svt_pinned_alloc'ed: straight DMA
            HIP_TRY(hipMemcpyAsync(out, b->out_dev, n * sizeof(svt_result), hipMemcpyDeviceToHost, b->stream));
            HIP_TRY(hipStreamSynchronize(b->stream));
            return SVT_OK;
        }
        return d2h_staged(out, b->out_dev, n * sizeof(svt_result), b->stream);
    }
    // tagged 96-byte records: down through the pinned ring in pieces of whole records, every record put where its tag says
    // while the next piece is on the wire (that copy out of the ring slot is there for pageable memory anyway)
    StagingRing& ring = current_ring();
    std::lock_guard<std::mutex> guard(ring.lock);
    SVT_TRY(ring.ensure());
    const uint64_t per_piece = StagingRing::kPiece / sizeof(svt_result96), total = b->out_slots;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(b->out_dev);
    uint64_t s0 = 0, prev_n = 0;
    int slot = 0, prev_slot = -1;
    Placed placed(n);
    while (s0 < total || prev_slot >= 0) {
        uint64_t cnt = 0;
        if (s0 < total) {
            cnt = std::min(per_piece, total - s0);
            HIP_TRY(hipMemcpyAsync(ring.buf[slot], src + s0 * sizeof(svt_result96), cnt * sizeof(svt_result96), hipMemcpyDeviceToHost, b->stream));
        }
        if (prev_slot >= 0) expand96(static_cast<const svt_result96*>(ring.buf[prev_slot]), prev_n, out, placed);
        HIP_TRY(hipStreamSynchronize(b->stream));
        prev_slot = cnt ? slot : -1;
        prev_n = cnt;
        s0 += cnt;
        slot = (slot + 1) % 2;
    }
    if (!placed.covers(n)) return fail(SVT_ERR_INTERNAL, "the device result records do not cover every unit exactly once");
    return SVT_OK;
}

// payload_of(u) = first payload item (16 bytes each) of unit u; upload(i0, i1) enqueues items [i0, i1) on b->stream
template <typename PayloadOf, typename Upload>
int run_pipelined(svt_batch* b, svt_result* out, bool* download_left, PayloadOf&& payload_of, Upload&& upload)
{
    *download_left = false;
    const uint64_t n = b->n_units;
    StageTimer tm0;
    PipeStreams ps;
    SVT_TRY(g_handles.get_stream(&ps.compute));
    SVT_TRY(g_handles.get_stream(&ps.down));
    const bool r96 = (b->flags & SVT_FLAG_RESULT96) != 0;
    const bool out_pinned = n && !r96 && g_pinned.is_pinned(out, n * sizeof(svt_result));
    // 96-byte device records: every piece comes down into a page-locked scratch as soon as its launch is through and is
    // expanded into the caller's array while the later pieces are still on their way
    struct Scratch { void* p = nullptr; ~Scratch() { g_pinned.put(p); } } scratch;
    struct Piece { uint64_t u0, u1, s0, s1; hipEvent_t down; };
    std::vector<Piece> pieces;
    StageTimer tm;
    static const uint64_t piece_mb = std::getenv("SVT_PIPE_MB") ? std::strtoull(std::getenv("SVT_PIPE_MB"), nullptr, 10) : 64;
    const uint64_t kPieceItems = (std::max<uint64_t>(piece_mb, 1) << 20) / 16;   // payload per piece (the staging ring's piece size)
    // the pieces: whole units up to kPieceItems of payload each (at least one unit); their tagged result records
    // (SVT_FLAG_RESULT96) take whole workgroups' worth of slots per launch
    uint64_t total_slots = 0;
    for (uint64_t u0 = 0; u0 < n;) {
        uint64_t lo = u0 + 1, hi = n;
        const uint64_t want = payload_of(u0) + kPieceItems;
        while (lo < hi) {   // largest u1 with payload_of(u1) <= want
            const uint64_t mid = lo + (hi - lo + 1) / 2;
            if (payload_of(mid) <= want) lo = mid; else hi = mid - 1;
        }
        const uint64_t slots = r96 ? slots_of_launch(b, lo - u0) : lo - u0;
        pieces.push_back(Piece{u0, lo, total_slots, total_slots + slots, nullptr});
        total_slots += slots;
        u0 = lo;
    }
    if (r96 && n) {
        if (total_slots >= 0xFFFFFFF0ull) return fail(SVT_ERR_INVALID, "too many result slots in one batch");
        SVT_TRY(ensure_result_slots(b, total_slots));
        b->out_slots = total_slots;
        scratch.p = g_pinned.get(total_slots * sizeof(svt_result96));
        if (!scratch.p) return fail(SVT_ERR_NOMEM, "page-locked scratch for the result records");
    }
    for (Piece& pc : pieces) {
        const uint64_t u0 = pc.u0, u1 = pc.u1;
        SVT_TRY(upload(payload_of(u0), payload_of(u1)));
        hipEvent_t landed, done;
        SVT_TRY(ps.event(&landed));
        HIP_TRY(hipEventRecord(landed, b->stream));
        HIP_TRY(hipStreamWaitEvent(ps.compute, landed, 0));
        SVT_TRY(launch_range(b, u0, u1, ps.compute, pc.s0));
        if (out_pinned) {
            SVT_TRY(ps.event(&done));
            HIP_TRY(hipEventRecord(done, ps.compute));
            HIP_TRY(hipStreamWaitEvent(ps.down, done, 0));
            HIP_TRY(hipMemcpyAsync(out + u0, b->out_dev + u0, (u1 - u0) * sizeof(svt_result), hipMemcpyDeviceToHost, ps.down));
        } else if (r96) {
            SVT_TRY(ps.event(&done));
            HIP_TRY(hipEventRecord(done, ps.compute));
            HIP_TRY(hipStreamWaitEvent(ps.down, done, 0));
            HIP_TRY(hipMemcpyAsync(static_cast<unsigned char*>(scratch.p) + pc.s0 * sizeof(svt_result96),
                                   reinterpret_cast<const unsigned char*>(b->out_dev) + pc.s0 * sizeof(svt_result96),
                                   (pc.s1 - pc.s0) * sizeof(svt_result96), hipMemcpyDeviceToHost, ps.down));
            SVT_TRY(ps.event(&pc.down));
            HIP_TRY(hipEventRecord(pc.down, ps.down));
        }
    }
    tm.mark("pipeline: pieces enqueued");
    // (96-byte records: piece k is expanded as soon as it is down, while the later pieces are still going up; should the pass
    // report a contract violation below, what was expanded is discarded with the error)
    Placed placed(n);
    if (r96)
        for (const Piece& pc : pieces) {
            HIP_TRY(hipEventSynchronize(pc.down));
            expand96(static_cast<const svt_result96*>(scratch.p) + pc.s0, pc.s1 - pc.s0, out, placed);
        }
    HIP_TRY(hipStreamSynchronize(b->stream));
    tm.mark("pipeline: uploads done");
    HIP_TRY(hipStreamSynchronize(ps.compute));
    b->have_results = true;
    SVT_TRY(check_stream_errors(b));
    tm.mark("pipeline: passes done");
    if (r96) {
        if (!placed.covers(n)) return fail(SVT_ERR_INTERNAL, "the device result records do not cover every unit exactly once");
    } else if (out_pinned) {
        HIP_TRY(hipStreamSynchronize(ps.down));
    } else {
        *download_left = true;   // pageable output: the caller downloads through the staging ring once it is free
    }
    tm.mark("pipeline: downloads done");
    (void)tm0;
    return SVT_OK;
}

constexpr uint64_t kPipelineMinUnits = 32768;   // below this one upload + one launch is as good

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int svt_version(void) { return SVT_ABI_VERSION; }

int svt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* svt_last_error(void) { return g_err.c_str(); }

static int svt_batch_create_impl(const svt_evidence_batch* in, int device, unsigned flags, svt_batch** out)
{
    if (!in || !out) return fail(SVT_ERR_INVALID, "null argument");
    *out = nullptr;
    const uint64_t n = in->n_units;
    if (flags & ~kKnownFlags) return fail(SVT_ERR_INVALID, "unknown flag bits");
    if (n >= 0xFFFFFFF0ull) return fail(SVT_ERR_INVALID, "too many units in one batch (< 2^32)");
    if (in->n_libs == 0 || in->n_libs > 256 || !in->libs) return fail(SVT_ERR_INVALID, "n_libs must be 1..256");
    if (n && (!in->rec_offset || !in->units)) return fail(SVT_ERR_INVALID, "null unit arrays");
    if (n && in->rec_offset[0] != 0) return fail(SVT_ERR_INVALID, "rec_offset[0] must be 0");
    if (n && in->rec_offset[n] && !in->records) return fail(SVT_ERR_INVALID, "null records");
    if (!(in->split_weight >= 0.0) || !(in->disc_weight >= 0.0) || !std::isfinite(in->split_weight) ||
        !std::isfinite(in->disc_weight))
        return fail(SVT_ERR_INVALID, "weights must be finite and >= 0");

    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    svt_batch* b = new (std::nothrow) svt_batch();
    if (!b) return fail(SVT_ERR_NOMEM, "out of host memory");
    b->device = device;
    b->flags = flags;
    b->layout = kLayoutStream;
    b->n_units = n;
    b->n_records = n ? in->rec_offset[n] : 0;
    const int rc = create_stream(in, b);
    if (rc != SVT_OK) {
        const std::string keep = g_err;
        free_batch(b);
        g_err = keep;
        return rc;
    }
    *out = b;
    return SVT_OK;
}

int svt_batch_create(const svt_evidence_batch* in, int device, unsigned flags, svt_batch** out)
{
    return guarded([&] { return svt_batch_create_impl(in, device, flags, out); });
}

// svt_batch_create with the records in pieces (include/svtyper_hip.h): create_stream leaves the record upload to this function
// (as it does for the pipelined one-shot), every segment goes through the staging ring to its place in the device array.  A
// batch whose library windows have to be read off the records (several libraries, units without hints) needs the records
// while it is created: its segments are put together in page-locked scratch first -- the rare case.
static int svt_batch_create_segments_impl(const svt_evidence_batch* in, const svt_record_segment* segments, uint32_t n_segments,
                                          int device, unsigned flags, svt_batch** out)
{
    if (!in || !out || (n_segments && !segments)) return fail(SVT_ERR_INVALID, "null argument");
    *out = nullptr;
    const uint64_t n = in->n_units;
    if (n && !in->rec_offset) return fail(SVT_ERR_INVALID, "null unit arrays");
    const uint64_t n_rec = n ? in->rec_offset[n] : 0;
    uint64_t have = 0;
    for (uint32_t k = 0; k < n_segments; ++k) {
        if (segments[k].n_records && !segments[k].records) return fail(SVT_ERR_INVALID, "svt_batch_create_segments: null segment");
        if (segments[k].n_records > n_rec - have) return fail(SVT_ERR_INVALID, "svt_batch_create_segments: the segments hold more records than rec_offset[n_units]");
        have += segments[k].n_records;
    }
    if (have != n_rec) return fail(SVT_ERR_INVALID, "svt_batch_create_segments: the segments hold fewer records than rec_offset[n_units]");
    svt_evidence_batch eb = *in;
    eb.records = nullptr;
    bool hinted = true;      // (create_stream's rule: the windows come from the hints only when every unit has one)
    if (in->n_libs > 1 && in->units && !(flags & SVT_FLAG_GENERAL_TABLES))
        for (uint64_t u = 0; u < n && hinted; ++u) hinted = ((in->units[u].libs >> 8) & 0xffu) != 0u;
    if (!hinted || (flags & ~kKnownFlags) || n_rec == 0) {
        struct Scratch { void* p = nullptr; ~Scratch() { g_pinned.put(p); } } scratch;
        if (n_rec) {
            scratch.p = g_pinned.get(n_rec * sizeof(svt_record));
            if (!scratch.p) return fail(SVT_ERR_NOMEM, "out of page-locked host memory");
            char* at = static_cast<char*>(scratch.p);
            for (uint32_t k = 0; k < n_segments; ++k) {
                std::memcpy(at, segments[k].records, segments[k].n_records * sizeof(svt_record));
                at += segments[k].n_records * sizeof(svt_record);
            }
            eb.records = static_cast<const svt_record*>(scratch.p);
        }
        return svt_batch_create_impl(&eb, device, flags, out);
    }
    // the checks of svt_batch_create_impl (the records are not looked at on the host: the pass itself checks their contract)
    if (n >= 0xFFFFFFF0ull) return fail(SVT_ERR_INVALID, "too many units in one batch (< 2^32)");
    if (in->n_libs == 0 || in->n_libs > 256 || !in->libs) return fail(SVT_ERR_INVALID, "n_libs must be 1..256");
    if (n && !in->units) return fail(SVT_ERR_INVALID, "null unit arrays");
    if (n && in->rec_offset[0] != 0) return fail(SVT_ERR_INVALID, "rec_offset[0] must be 0");
    if (!(in->split_weight >= 0.0) || !(in->disc_weight >= 0.0) || !std::isfinite(in->split_weight) || !std::isfinite(in->disc_weight))
        return fail(SVT_ERR_INVALID, "weights must be finite and >= 0");
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    svt_batch* b = new (std::nothrow) svt_batch();
    if (!b) return fail(SVT_ERR_NOMEM, "out of host memory");
    b->device = device;
    b->flags = flags;
    b->layout = kLayoutStream;
    b->n_units = n;
    b->n_records = n_rec;
    int rc = create_stream(&eb, b, nullptr, 0, /*defer_records=*/true);
    if (rc == SVT_OK && b->records_resident) rc = fail(SVT_ERR_INTERNAL, "svt_batch_create_segments: create_stream wanted the records");
    if (rc == SVT_OK) {
        Stager st(b->stream);
        char* at = static_cast<char*>(b->d_records);
        for (uint32_t k = 0; k < n_segments && rc == SVT_OK; ++k) {
            rc = st.copy(at, segments[k].records, segments[k].n_records * sizeof(svt_record));
            at += segments[k].n_records * sizeof(svt_record);
        }
        if (rc == SVT_OK) rc = st.finish();
        if (rc == SVT_OK) b->records_resident = true;
    }
    if (rc != SVT_OK) {
        const std::string keep = g_err;
        free_batch(b);
        g_err = keep;
        return rc;
    }
    *out = b;
    return SVT_OK;
}

int svt_batch_create_segments(const svt_evidence_batch* in, const svt_record_segment* segments, uint32_t n_segments, int device,
                              unsigned flags, svt_batch** out)
{
    return guarded([&] { return svt_batch_create_segments_impl(in, segments, n_segments, device, flags, out); });
}

static int svt_batch_create_from_fragments_impl(const svt_fragment_batch* in, int device, unsigned flags,
                                    svt_record* records_out, svt_batch** out)
{
    if (!in || !out) return fail(SVT_ERR_INVALID, "null argument");
    *out = nullptr;
    const uint64_t n = in->n_units;
    if (flags & ~kKnownFlags) return fail(SVT_ERR_INVALID, "unknown flag bits");
    if (n >= 0xFFFFFFF0ull) return fail(SVT_ERR_INVALID, "too many units in one batch (< 2^32)");
    if (in->n_libs == 0 || in->n_libs > 256 || !in->libs) return fail(SVT_ERR_INVALID, "n_libs must be 1..256");
    if (n && (!in->frag_offset || !in->breakpoints)) return fail(SVT_ERR_INVALID, "null unit arrays");
    if (n && in->frag_offset[0] != 0) return fail(SVT_ERR_INVALID, "frag_offset[0] must be 0");
    const uint64_t n_frag = n ? in->frag_offset[n] : 0;
    if (n_frag && !in->fragments) return fail(SVT_ERR_INVALID, "null fragments");
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    // unit headers
    std::vector<svt_unit> units(n);
    for (uint64_t u = 0; u < n; ++u) {
        const svt_breakpoint& bp = in->breakpoints[u];
        if (in->frag_offset[u + 1] < in->frag_offset[u]) return fail(SVT_ERR_INVALID, "frag_offset not monotone");
        if (bp.svtype > SVT_SVTYPE_BND) return fail(SVT_ERR_INVALID, "bad svtype");
        svt_unit U{};
        U.var_length = bp.svtype == SVT_SVTYPE_DEL ? bp.var_length : 0;
        const int64_t delta = (int64_t)bp.pos_b - (int64_t)bp.pos_a;            // classic.py:339
        U.pos_delta = (int32_t)std::max<int64_t>(INT32_MIN, std::min<int64_t>(INT32_MAX, delta));
        U.sample = bp.sample;
        U.svtype = bp.svtype;
        U.flags = (bp.flags & SVT_BP_SKIP) ? SVT_UNIT_SKIP : 0;
        U.libs = bp.reserved[0] & 0xffffu;      // SVT_UNIT_LIBS hint of the unit's sample
        units[u] = U;
    }
    // library descriptors (the flank of is_pair_straddle is lib.mean + lib.sd * 3)
    std::vector<LibDesc> libs(in->n_libs);
    for (uint32_t l = 0; l < in->n_libs; ++l) {
        if (!std::isfinite(in->libs[l].mean) || !std::isfinite(in->libs[l].sd)) return fail(SVT_ERR_INVALID, "library moments not finite");
        libs[l].v_nondel = in->libs[l].mean + in->libs[l].sd * 3;
    }

    // geometry on the device
    hipStream_t s = nullptr;
    SVT_TRY(g_handles.get_stream(&s));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); g_handles.put_stream(s); } } sg{s};
    StageTimer tm;
    // the two big buffers of this stage come from the pool svt_batch_destroy refills (svt_host_transfer.h)
    struct Pooled {
        int device;
        void* p = nullptr;
        uint64_t cap = 0;
        ~Pooled() { g_pool.put(device, p, cap); }
        int get(uint64_t bytes, bool records = false) { return g_pool.get(device, bytes, &p, &cap, records); }
        void* release() { void* q = p; p = nullptr; return q; }
    } d_frags{device}, d_records{device};
    DevScratch d_frag_off, d_bps, d_libs, d_err;
    {
        Stager st(s);
        SVT_TRY(d_frags.get(n_frag * sizeof(svt_fragment)));
        SVT_TRY(st.copy(d_frags.p, in->fragments, n_frag * sizeof(svt_fragment)));
        SVT_TRY(d_frag_off.alloc((n + 1) * sizeof(uint64_t)));
        if (n) SVT_TRY(st.copy(d_frag_off.p, in->frag_offset, (n + 1) * sizeof(uint64_t)));
        SVT_TRY(d_bps.alloc(n * sizeof(svt_breakpoint)));
        SVT_TRY(st.copy(d_bps.p, in->breakpoints, n * sizeof(svt_breakpoint)));
        SVT_TRY(upload(d_libs, libs, st));
        SVT_TRY(st.finish());
        tm.mark("H2D fragment summaries + unit arrays (staged)");
    }
    SVT_TRY(d_records.get((n_frag + kBlockRecords) * sizeof(uint4), /*records=*/true));   // whole 128-byte blocks (kLayoutStream)
    SVT_TRY(d_err.alloc(sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), s));
    if (n_frag) {
        GeomArgs g{};
        g.frags = static_cast<const uint4*>(d_frags.p);
        g.frag_offset = d_frag_off.as<uint64_t>();
        g.n_units = n;
        g.bps = d_bps.as<svt_breakpoint>();
        g.libs = d_libs.as<LibDesc>();
        g.n_frags = n_frag;
        g.n_libs = in->n_libs;
        g.min_aligned = in->min_aligned;
        g.split_slop = in->split_slop;
        g.records = static_cast<uint4*>(d_records.p);
        g.err = d_err.as<uint32_t>();
        hipLaunchKernelGGL(svt_geometry_kernel, dim3((unsigned)((n_frag + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, g);
        HIP_TRY(hipGetLastError());
    }
    uint32_t err_bits = 0;
    HIP_TRY(hipMemcpyAsync(&err_bits, d_err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (records_out && n_frag) SVT_TRY(d2h_staged(records_out, d_records.p, n_frag * sizeof(uint4), s));
    HIP_TRY(hipStreamSynchronize(s));
    tm.mark("geometry kernel (+ copies)");
    if (err_bits) return fail(SVT_ERR_INVALID, "invalid fragment summaries: library index >= n_libs");

    // the resident batch, from the records that are already in HBM
    svt_evidence_batch eb{};
    eb.n_units = n;
    eb.rec_offset = in->frag_offset;
    eb.units = units.data();
    eb.records = nullptr;
    eb.n_libs = in->n_libs;
    eb.libs = in->libs;
    eb.split_weight = in->split_weight;
    eb.disc_weight = in->disc_weight;
    if (!(eb.split_weight >= 0.0) || !(eb.disc_weight >= 0.0) || !std::isfinite(eb.split_weight) ||
        !std::isfinite(eb.disc_weight))
        return fail(SVT_ERR_INVALID, "weights must be finite and >= 0");
    svt_batch* b = new (std::nothrow) svt_batch();
    if (!b) return fail(SVT_ERR_NOMEM, "out of host memory");
    b->device = device;
    b->flags = flags;
    b->layout = kLayoutStream;
    b->n_units = n;
    b->n_records = n_frag;
    const uint64_t cap = d_records.cap;
    const int rc = create_stream(&eb, b, d_records.p, cap);
    if (b->d_records == d_records.p) d_records.release();   // the batch owns the records now
    if (rc != SVT_OK) {
        const std::string keep = g_err;
        free_batch(b);
        g_err = keep;
        return rc;
    }
    *out = b;
    return SVT_OK;
}

int svt_batch_create_from_fragments(const svt_fragment_batch* in, int device, unsigned flags, svt_record* records_out, svt_batch** out)
{
    return guarded([&] { return svt_batch_create_from_fragments_impl(in, device, flags, records_out, out); });
}

static int svt_batch_genotype_impl(svt_batch* b, int sync)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    HIP_TRY(hipSetDevice(b->device));
    SVT_TRY(launch_genotype(b));
    b->have_results = true;
    if (sync) {
        HIP_TRY(hipStreamSynchronize(b->stream));
        SVT_TRY(check_stream_errors(b));
    }
    return SVT_OK;
}

int svt_batch_genotype(svt_batch* b, int sync)
{
    return guarded([&] { return svt_batch_genotype_impl(b, sync); });
}

static int svt_batch_genotype_n_impl(svt_batch* b, int iters)
{
    if (!b || iters <= 0) return fail(SVT_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    for (int i = 0; i < iters; ++i) SVT_TRY(launch_genotype(b));
    b->have_results = true;
    return SVT_OK;
}

int svt_batch_genotype_n(svt_batch* b, int iters)
{
    return guarded([&] { return svt_batch_genotype_n_impl(b, iters); });
}

int svt_batch_sync(svt_batch* b)
{
    return guarded([&]() -> int {
        if (!b) return fail(SVT_ERR_INVALID, "null batch");
        HIP_TRY(hipSetDevice(b->device));
        HIP_TRY(hipStreamSynchronize(b->stream));
        return check_stream_errors(b);
    });
}

static int svt_batch_genotype_timed_impl(svt_batch* b, int iters, float* ms_total)
{
    if (!b || !ms_total || iters <= 0) return fail(SVT_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipEventRecord(b->ev0, b->stream));
    for (int i = 0; i < iters; ++i) SVT_TRY(launch_genotype(b));
    HIP_TRY(hipEventRecord(b->ev1, b->stream));
    HIP_TRY(hipEventSynchronize(b->ev1));
    HIP_TRY(hipEventElapsedTime(ms_total, b->ev0, b->ev1));
    b->have_results = true;
    return check_stream_errors(b);
}

// svt_batch_tune_placement (include/svtyper_hip.h): audition device buffers for the result records and for the records.
// Which physical blocks of HBM the two big buffers of a batch lie in moves the pass by up to 8 % (levels, stable
// for the life of an allocation, that nothing at allocation time predicts): with 288 GB of HBM the cheap answer is to allocate a
// handful of candidates, run the REAL pass over each once the clocks are up, keep the fastest and hand the others back.  The
// kept buffers return to the pool with the batch, so the batches of a chunked run that follow inherit them.
static int svt_batch_tune_placement_impl(svt_batch* b, int result_candidates, int record_candidates, float* before_ms, float* after_ms)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    if (result_candidates < 0 || record_candidates < 0 || result_candidates > 64 || record_candidates > 16) return fail(SVT_ERR_INVALID, "0..64 result and 0..16 record candidates");
    if (before_ms) *before_ms = 0.f;
    if (after_ms) *after_ms = 0.f;
    if (b->n_units == 0) return SVT_OK;
    if (b->out_dev != b->d_out) return fail(SVT_ERR_INVALID, "the result records are bound to a caller's buffer (svt_batch_bind_device_results)");
    HIP_TRY(hipSetDevice(b->device));
    // Whatever way this function is left -- an audition cut short by a failing launch or copy included -- the pass's arguments
    // point at the batch's OWN buffers again (the candidate guards below synchronise the stream before they release anything),
    // and a batch left half way has no results.
    struct Restore {
        svt_batch* b;
        bool done = false;
        ~Restore()
        {
            if (!done) {
                (void)hipStreamSynchronize(b->stream);
                b->have_results = false;
            }
            b->out_dev = b->d_out;
            b->sargs.out = b->pargs.out = b->d_out;
            if (b->layout == kLayoutStream && b->records_resident) b->sargs.records = static_cast<const uint4*>(b->d_records);
        }
    } restore{b};
    auto pass_ms = [&](int iters, float* ms) -> int {      // `iters` back-to-back launches, per launch
        float total = 0.f;
        HIP_TRY(hipEventRecord(b->ev0, b->stream));
        for (int i = 0; i < iters; ++i) SVT_TRY(launch_genotype(b));
        HIP_TRY(hipEventRecord(b->ev1, b->stream));
        HIP_TRY(hipEventSynchronize(b->ev1));
        HIP_TRY(hipEventElapsedTime(&total, b->ev0, b->ev1));
        *ms = total / (float)iters;
        return SVT_OK;
    };
    auto best_of = [&](int groups, int iters, float* ms) -> int {
        float best = 0.f;
        for (int g = 0; g < groups; ++g) {
            float t = 0.f;
            SVT_TRY(pass_ms(iters, &t));
            if (g == 0 || t < best) best = t;
        }
        *ms = best;
        return SVT_OK;
    };
    // clocks up: ~40 ms of passes (a device that idled runs its first launches 5-8 % slow)
    {
        float one = 0.f;
        SVT_TRY(pass_ms(2, &one));
        const int n = (int)std::min(400.0, std::max(4.0, 40.0 / std::max(one, 0.01f)));
        SVT_TRY(pass_ms(n, &one));
    }
    float current = 0.f;
    SVT_TRY(best_of(3, 10, &current));
    if (before_ms) *before_ms = current;
    const bool resident_records = b->layout == kLayoutStream && b->records_resident && b->sargs.records == static_cast<const uint4*>(b->d_records);
    struct Cand { void* p; uint64_t cap; float ms; };
    // The whole audition stays within ~0.3 s of uninterrupted passes: beyond that the device alternates between its level and one
    // ~4 % slower until it has idled (profiles/r04_placement_tuning.txt), and candidates measured in that state are ranked by the
    // state, not by their placement (an audition of 48 + 12 candidates at ten launches each kept a 0.301 ms pair where 32 + 8 found
    // 0.285 twice).  So the launches per measurement follow from the pass time and the number of candidates.
    const int n_cand = result_candidates + (resident_records ? record_candidates : 0);
    const int iters = std::max(3, std::min(10, (int)(250.0f / (float)std::max(n_cand, 1) / (2.0f * std::max(current, 1e-3f)))));
    // ---- result records: plain allocations (they may be handed to RCCL or to another process)
    if (result_candidates > 0) {
        const uint64_t bytes = std::max<uint64_t>(b->cap_out, std::max<uint64_t>(b->out_slots, 1) * result_bytes(b));
        std::vector<Cand> cands;
        struct FreeAll { std::vector<Cand>& c; hipStream_t s; ~FreeAll() { (void)hipStreamSynchronize(s); for (Cand& x : c) if (x.p) (void)hipFree(x.p); } } guard{cands, b->stream};
        for (int i = 0; i < result_candidates; ++i) {
            void* p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }     // (out of memory: audition what there is)
            cands.push_back(Cand{p, bytes, 0.f});
        }
        size_t best = cands.size();
        for (size_t i = 0; i < cands.size(); ++i) {
            b->sargs.out = b->pargs.out = static_cast<svt_result*>(cands[i].p);
            SVT_TRY(best_of(2, iters, &cands[i].ms));
            if (best == cands.size() || cands[i].ms < cands[best].ms) best = i;
        }
        if (best != cands.size()) {      // the winner once more, against the incumbent measured the same way (a single fast group is not a level)
            b->sargs.out = b->pargs.out = static_cast<svt_result*>(cands[best].p);
            SVT_TRY(best_of(3, iters, &cands[best].ms));
        }
        if (best != cands.size() && cands[best].ms < current * 0.995f) {
            HIP_TRY(hipStreamSynchronize(b->stream));
            g_pool.put(b->device, b->d_out, b->cap_out);
            b->d_out = static_cast<svt_result*>(cands[best].p);
            b->cap_out = cands[best].cap;
            current = cands[best].ms;
            cands[best].p = nullptr;
        }
        b->out_dev = b->d_out;
        b->sargs.out = b->pargs.out = b->d_out;
    }
    // ---- records (canonical records resident in the batch's own buffer): candidates of the pool's own kind, filled by device copies
    if (record_candidates > 0 && resident_records) {
        const uint64_t bytes = ((uint64_t)b->sargs.last_blk + 1) * 128;      // the records as the kernel reads them: whole 128-byte blocks
        std::vector<Cand> cands;
        struct FreeAll { std::vector<Cand>& c; int device; hipStream_t s; ~FreeAll() { (void)hipStreamSynchronize(s); for (Cand& x : c) if (x.p) g_pool.release(x.p, device); } } guard{cands, b->device, b->stream};
        for (int i = 0; i < record_candidates; ++i) {
            void* p = nullptr;
            uint64_t cap = 0;
            // (not from the pool's idle list: a buffer that sits there was this batch's neighbour in time, not a new draw)
            if (!(bytes + bytes / 8 >= DevicePool::kChunkedMin && g_pool.chunked_available && g_pool.alloc_chunked(b->device, bytes + bytes / 8, &p, &cap))) {
                if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
                cap = bytes;
            }
            cands.push_back(Cand{p, cap, 0.f});
            HIP_TRY(hipMemcpyAsync(p, b->d_records, bytes, hipMemcpyDeviceToDevice, b->stream));
        }
        size_t best = cands.size();
        for (size_t i = 0; i < cands.size(); ++i) {
            b->sargs.records = static_cast<const uint4*>(cands[i].p);
            SVT_TRY(best_of(2, iters, &cands[i].ms));
            if (best == cands.size() || cands[i].ms < cands[best].ms) best = i;
        }
        if (best != cands.size()) {
            b->sargs.records = static_cast<const uint4*>(cands[best].p);
            SVT_TRY(best_of(3, iters, &cands[best].ms));
        }
        if (best != cands.size() && cands[best].ms < current * 0.995f) {
            HIP_TRY(hipStreamSynchronize(b->stream));
            g_pool.put(b->device, b->d_records, b->cap_records);
            b->d_records = cands[best].p;
            b->cap_records = cands[best].cap;
            current = cands[best].ms;
            cands[best].p = nullptr;
        }
        b->sargs.records = static_cast<const uint4*>(b->d_records);
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (after_ms) *after_ms = current;
    b->have_results = true;      // (the last pass ran over the kept buffers)
    restore.done = true;
    return check_stream_errors(b);
}

int svt_batch_tune_placement(svt_batch* b, int result_candidates, int record_candidates, float* before_ms, float* after_ms)
{
    return guarded([&] { return svt_batch_tune_placement_impl(b, result_candidates, record_candidates, before_ms, after_ms); });
}

int svt_batch_genotype_timed(svt_batch* b, int iters, float* ms_total)
{
    return guarded([&] { return svt_batch_genotype_timed_impl(b, iters, ms_total); });
}

static int svt_batch_results_impl(svt_batch* b, svt_result* out, uint64_t n_units)
{
    if (!b || (!out && n_units)) return fail(SVT_ERR_INVALID, "null argument");
    if (!b->have_results) return fail(SVT_ERR_STATE, "svt_batch_genotype has not run");
    if (n_units != b->n_units) return fail(SVT_ERR_INVALID, "results n_units mismatch");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));   // the pass that produced the records
    SVT_TRY(check_stream_errors(b));
    return d2h_results(b, out);
}

int svt_batch_results(svt_batch* b, svt_result* out, uint64_t n_units)
{
    return guarded([&] { return svt_batch_results_impl(b, out, n_units); });
}

uint32_t svt_batch_result_bytes(const svt_batch* b) { return b ? result_bytes(b) : 0u; }

uint64_t svt_batch_result_slots(const svt_batch* b) { return b ? b->out_slots : 0; }

int svt_results_expand96(const svt_result96* in, uint64_t n_records, svt_result* out, uint64_t n_units)
{
    return guarded([&]() -> int {
        if ((n_records && !in) || (n_units && !out)) return fail(SVT_ERR_INVALID, "null argument");
        Placed placed(n_units);
        expand96(in, n_records, out, placed);
        if (placed.bad) return fail(SVT_ERR_INVALID, "svt_results_expand96: a record's unit is beyond n_units");
        if (!placed.covers(n_units)) return fail(SVT_ERR_INVALID, "svt_results_expand96: the records do not cover every unit exactly once");
        return SVT_OK;
    });
}

int svt_batch_result_order(svt_batch* b, uint32_t n_samples)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    if (b->layout != kLayoutStream) return fail(SVT_ERR_INVALID, "svt_batch_result_order: canonical records only (not packed evidence)");
    if (n_samples <= 1) {
        if (b->sargs.out_samples > 1) b->have_results = false;   // (site-major records are not results in unit order)
        b->sargs.out_samples = 0;
        b->sargs.out_sites = 0;
        return SVT_OK;
    }
    if (b->n_units % n_samples) return fail(SVT_ERR_INVALID, "svt_batch_result_order: n_units is not a multiple of n_samples");
    if (b->sargs.out_samples != n_samples) b->have_results = false;   // (records written in another order are not results of this one)
    b->sargs.out_samples = n_samples;
    b->sargs.out_sites = (uint32_t)(b->n_units / n_samples);
    return SVT_OK;
}

int svt_batch_device_results(svt_batch* b, svt_result** dev)
{
    if (!b || !dev) return fail(SVT_ERR_INVALID, "null argument");
    *dev = b->out_dev;
    return SVT_OK;
}

static int svt_batch_bind_device_results_impl(svt_batch* b, svt_result* dev, uint64_t capacity_bytes, bool have_capacity)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    if (dev && (reinterpret_cast<uintptr_t>(dev) & 127u)) return fail(SVT_ERR_INVALID, "result buffer must be 128-byte aligned");
    const uint64_t need = std::max<uint64_t>(b->out_slots, 1) * result_bytes(b);
    if (dev && have_capacity && capacity_bytes < need)
        return fail(SVT_ERR_INVALID, "result buffer too small: svt_batch_result_slots(b) * svt_batch_result_bytes(b) = " + std::to_string(need) + " bytes");
    b->out_dev = dev ? dev : b->d_out;
    b->sargs.out = b->out_dev;
    b->pargs.out = b->out_dev;
    b->bound_slots = dev ? (have_capacity ? capacity_bytes / result_bytes(b) : b->out_slots) : 0;   // what a later pass may write
    b->have_results = false;

    return SVT_OK;
}

int svt_batch_bind_device_results(svt_batch* b, svt_result* dev)
{
    return guarded([&] { return svt_batch_bind_device_results_impl(b, dev, 0, false); });
}

int svt_batch_bind_device_results2(svt_batch* b, void* dev, uint64_t capacity_bytes)
{
    return guarded([&] { return svt_batch_bind_device_results_impl(b, static_cast<svt_result*>(dev), capacity_bytes, true); });
}

int svt_batch_bytes(const svt_batch* b, uint64_t* algorithmic, uint64_t* resident)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    if (algorithmic) *algorithmic = 16 * b->n_records + (16 + 96) * b->n_units;
    if (resident) *resident = b->layout == kLayoutPacked ? 16 * b->n_slots + (12 + 16) * b->n_units
                                                         : 16 * b->n_records + (8 + 16) * b->n_units;
    return SVT_OK;
}

int svt_batch_layout(const svt_batch* b, int* compact, int* table_mode)
{
    if (!b) return fail(SVT_ERR_INVALID, "null batch");
    if (compact) *compact = b->layout;
    if (table_mode) *table_mode = b->mode;
    return SVT_OK;
}

static int svt_batch_site_qual_impl(svt_batch* b, uint32_t n_samples, const double* initial, double* qual_out, uint64_t n_sites)
{
    if (!b || (!qual_out && n_sites)) return fail(SVT_ERR_INVALID, "null argument");
    if (!b->have_results) return fail(SVT_ERR_STATE, "svt_batch_genotype has not run");
    if (n_samples == 0 || n_sites * n_samples != b->n_units) return fail(SVT_ERR_INVALID, "n_sites * n_samples != n_units");
    // the records of a sample-major batch were written site-major for out_samples samples per site: QUAL over groups of
    // another size would silently sum the wrong records
    if (b->layout == kLayoutStream && b->sargs.out_samples > 1 && n_samples != b->sargs.out_samples)
        return fail(SVT_ERR_INVALID, "svt_batch_site_qual: n_samples differs from the batch's svt_batch_result_order");
    if (n_sites == 0) return SVT_OK;
    HIP_TRY(hipSetDevice(b->device));
    SVT_TRY(check_stream_errors(b));   // (after svt_batch_genotype(b, 0) / _n nobody has looked at the contract word yet)
    DevScratch d_init, d_qual, d_entries, d_flag;
    SVT_TRY(d_qual.alloc(n_sites * sizeof(double)));
    if (initial) {
        SVT_TRY(d_init.alloc(n_sites * sizeof(double)));
        Stager st(b->stream);
        SVT_TRY(st.copy(d_init.p, initial, n_sites * sizeof(double)));
        SVT_TRY(st.finish());
    }
    if (b->flags & SVT_FLAG_RESULT96) {
        // tagged records lie in the kernel's order, not site by site: SQ and GT of every slot go where its tag says (16 bytes per
        // unit of device scratch), then the same running sum over a site's entries (svt_bayes_kernel.h) -- nothing but the
        // QUAL values crosses PCIe (this used to bring every record down: 2 GB and 0.6 s for the 16 M units of configs[4])
        SVT_TRY(d_entries.alloc(b->n_units * sizeof(QualEntry)));
        SVT_TRY(d_flag.alloc(sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(d_flag.p, 0, sizeof(uint32_t), b->stream));
        hipLaunchKernelGGL(svt_site_qual_scatter_kernel, dim3((unsigned)((b->out_slots + kBlock - 1) / kBlock)), dim3(kBlock), 0, b->stream,
                           reinterpret_cast<const svt_result96*>(b->out_dev), b->out_slots, b->n_units, d_entries.as<QualEntry>(), d_flag.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(svt_site_qual_entries_kernel, dim3((unsigned)((n_sites + kBlock - 1) / kBlock)), dim3(kBlock), 0, b->stream,
                           d_entries.as<QualEntry>(), n_samples, initial ? d_init.as<double>() : nullptr, d_qual.as<double>(), n_sites);
        HIP_TRY(hipGetLastError());
        uint32_t bad = 0;
        SVT_TRY(d2h_staged(&bad, d_flag.p, sizeof bad, b->stream));
        if (bad) return fail(SVT_ERR_INTERNAL, "svt_batch_site_qual: a result record carries a unit beyond the batch");
        return d2h_staged(qual_out, d_qual.p, n_sites * sizeof(double), b->stream);
    }
    hipLaunchKernelGGL(svt_site_qual_kernel, dim3((unsigned)((n_sites + kBlock - 1) / kBlock)), dim3(kBlock), 0, b->stream,
                       reinterpret_cast<const unsigned char*>(b->out_dev), (uint32_t)sizeof(svt_result), (uint32_t)offsetof(svt_result, gt), n_samples,
                       initial ? d_init.as<double>() : nullptr, d_qual.as<double>(), n_sites);
    HIP_TRY(hipGetLastError());
    return d2h_staged(qual_out, d_qual.p, n_sites * sizeof(double), b->stream);
}

int svt_batch_site_qual(svt_batch* b, uint32_t n_samples, const double* initial, double* qual_out, uint64_t n_sites)
{
    return guarded([&] { return svt_batch_site_qual_impl(b, n_samples, initial, qual_out, n_sites); });
}

static int svt_bayes_gt_impl(const int32_t* ref, const int32_t* alt, const uint8_t* is_dup, uint64_t n, double* out,
                 int device)
{
    if (n == 0) return SVT_OK;
    if (!ref || !alt || !is_dup || !out) return fail(SVT_ERR_INVALID, "null argument");
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    int64_t max_total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (ref[i] < 0 || alt[i] < 0) return fail(SVT_ERR_INVALID, "negative read count");
        max_total = std::max<int64_t>(max_total, (int64_t)ref[i] + alt[i]);
    }
    if (max_total >= (1 << 24)) return fail(SVT_ERR_INVALID, "ref + alt must be < 2^24");
    std::vector<double> l10((size_t)max_total + 2);
    l10[0] = 0.0;
    for (size_t i = 1; i < l10.size(); ++i) l10[i] = py_log10((double)i);
    GtConsts c{};
    fill_gt_consts(c, 1.0, 1.0);
    HIP_TRY(hipSetDevice(device));
    DevScratch d_ref, d_alt, d_dup, d_l10, d_out;
    SVT_TRY(d_ref.alloc(n * sizeof(int32_t)));
    SVT_TRY(d_alt.alloc(n * sizeof(int32_t)));
    SVT_TRY(d_dup.alloc(n));
    SVT_TRY(d_l10.alloc(l10.size() * sizeof(double)));
    SVT_TRY(d_out.alloc(n * 4 * sizeof(double)));
    {
        Stager st(nullptr);
        SVT_TRY(st.copy(d_ref.p, ref, n * sizeof(int32_t)));
        SVT_TRY(st.copy(d_alt.p, alt, n * sizeof(int32_t)));
        SVT_TRY(st.copy(d_dup.p, is_dup, n));
        SVT_TRY(st.copy(d_l10.p, l10.data(), l10.size() * sizeof(double)));
        SVT_TRY(st.finish());
    }
    hipLaunchKernelGGL(svt_bayes_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                       d_ref.as<int32_t>(), d_alt.as<int32_t>(), d_dup.as<uint8_t>(), n, d_l10.as<double>(), c,
                       d_out.as<double>());
    HIP_TRY(hipGetLastError());
    return d2h_staged(out, d_out.p, n * 4 * sizeof(double), nullptr);
}

int svt_bayes_gt(const int32_t* ref, const int32_t* alt, const uint8_t* is_dup, uint64_t n, double* out, int device)
{
    return guarded([&] { return svt_bayes_gt_impl(ref, alt, is_dup, n, out, device); });
}

static int svt_genotype_counts_impl(const double* counts, const uint8_t* is_dup, uint64_t n, double split_weight,
                                    double disc_weight, svt_result* out, int device)
{
    if (n == 0) return SVT_OK;
    if (!counts || !is_dup || !out) return fail(SVT_ERR_INVALID, "null argument");
    if (!(split_weight >= 0.0) || !(disc_weight >= 0.0) || !std::isfinite(split_weight) || !std::isfinite(disc_weight))
        return fail(SVT_ERR_INVALID, "weights must be finite and >= 0");
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    double max_total = 0.0;   // bound of QR + QA: the log10 table must reach it
    for (uint64_t i = 0; i < n; ++i) {
        const double* t = counts + 5 * i;
        for (int k = 0; k < 5; ++k)
            if (!(t[k] >= 0.0) || !std::isfinite(t[k])) return fail(SVT_ERR_INVALID, "counts must be finite and >= 0");
        max_total = std::max(max_total, split_weight * ((t[0] + t[1]) + t[2]) + disc_weight * (t[3] + t[4]));
    }
    if (max_total >= (double)(1 << 24)) return fail(SVT_ERR_INVALID, "weighted counts must stay below 2^24");
    std::vector<double> l10((size_t)max_total + 4);
    l10[0] = 0.0;
    for (size_t i = 1; i < l10.size(); ++i) l10[i] = py_log10((double)i);
    GtConsts c{};
    fill_gt_consts(c, split_weight, disc_weight);
    HIP_TRY(hipSetDevice(device));
    DevScratch d_counts, d_dup, d_l10, d_out;
    SVT_TRY(d_counts.alloc(n * 5 * sizeof(double)));
    SVT_TRY(d_dup.alloc(n));
    SVT_TRY(d_l10.alloc(l10.size() * sizeof(double)));
    SVT_TRY(d_out.alloc(n * sizeof(svt_result)));
    {
        Stager st(nullptr);
        SVT_TRY(st.copy(d_counts.p, counts, n * 5 * sizeof(double)));
        SVT_TRY(st.copy(d_dup.p, is_dup, n));
        SVT_TRY(st.copy(d_l10.p, l10.data(), l10.size() * sizeof(double)));
        SVT_TRY(st.finish());
    }
    hipLaunchKernelGGL(svt_counts_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0,
                       d_counts.as<double>(), d_dup.as<uint8_t>(), n, d_l10.as<double>(), c, d_out.as<svt_result>());
    HIP_TRY(hipGetLastError());
    return d2h_staged(out, d_out.p, n * sizeof(svt_result), nullptr);
}

int svt_genotype_counts(const double* counts, const uint8_t* is_dup, uint64_t n, double split_weight, double disc_weight,
                        svt_result* out, int device)
{
    return guarded([&] { return svt_genotype_counts_impl(counts, is_dup, n, split_weight, disc_weight, out, device); });
}

void* svt_pinned_alloc(size_t bytes)
{
    try { return g_pinned.get(bytes); } catch (...) { return nullptr; }
}

void svt_pinned_free(void* p)
{
    try { g_pinned.put(p); } catch (...) {}
}

int svt_pack_evidence(const svt_evidence_batch* in, svt_packed_evidence** out)
{
    return guarded([&] { return pack_evidence(in, out); });
}

void svt_packed_free(svt_packed_evidence* p)
{
    if (!p) return;
    delete reinterpret_cast<PackedOwner*>(p);   // `pub` is the owner's first member
}

static int svt_batch_create_packed_impl(const svt_packed_evidence* in, int device, unsigned flags, svt_batch** out)
{
    if (!in || !out) return fail(SVT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (flags & ~(SVT_FLAG_SSO_ASSOCIATION | SVT_FLAG_RESULT96)) return fail(SVT_ERR_INVALID, "packed evidence takes SVT_FLAG_SSO_ASSOCIATION and SVT_FLAG_RESULT96 only");
    if (in->n_units >= 0x55555550ull) return fail(SVT_ERR_INVALID, "too many units in one batch");
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    svt_batch* b = new (std::nothrow) svt_batch();
    if (!b) return fail(SVT_ERR_NOMEM, "out of host memory");
    b->device = device;
    b->flags = flags;
    b->layout = kLayoutPacked;
    b->n_units = in->n_units;
    b->n_records = in->n_records;
    const int rc = create_packed(in, b);
    if (rc != SVT_OK) {
        const std::string keep = g_err;
        free_batch(b);
        g_err = keep;
        return rc;
    }
    *out = b;
    return SVT_OK;
}

int svt_batch_create_packed(const svt_packed_evidence* in, int device, unsigned flags, svt_batch** out)
{
    return guarded([&] { return svt_batch_create_packed_impl(in, device, flags, out); });
}

static int svt_genotype_packed_impl(const svt_packed_evidence* in, svt_result* out, int device, unsigned flags)
{
    if (in && out && !(flags & ~(SVT_FLAG_SSO_ASSOCIATION | SVT_FLAG_RESULT96)) && in->n_units >= kPipelineMinUnits && in->n_units < 0x55555550ull &&
        in->slot_offset && in->slots) {
        const int ndev = svt_device_count();
        if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
        if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
        HIP_TRY(hipSetDevice(device));
        svt_batch* b = new (std::nothrow) svt_batch();
        if (!b) return fail(SVT_ERR_NOMEM, "out of host memory");
        b->device = device;
        b->flags = flags;
        b->layout = kLayoutPacked;
        b->n_units = in->n_units;
        b->n_records = in->n_records;
        int rc = create_packed(in, b, /*defer_slots=*/true);
        if (rc == SVT_OK) {
            bool download_left = false;
            {
            Stager st(b->stream);
            const bool pinned = g_pinned.is_pinned(in->slots, in->n_slots * 16);
            rc = run_pipelined(b, out, &download_left, [&](uint64_t u) { return (uint64_t)in->slot_offset[3 * u]; },
                               [&](uint64_t i0, uint64_t i1) -> int {
                                   char* dst = static_cast<char*>(b->d_records) + i0 * 16;
                                   const char* src = static_cast<const char*>(in->slots) + i0 * 16;
                                   if (pinned) { HIP_TRY(hipMemcpyAsync(dst, src, (i1 - i0) * 16, hipMemcpyHostToDevice, b->stream)); return SVT_OK; }
                                   return st.copy(dst, src, (i1 - i0) * 16);
                               });
            }
            if (rc == SVT_OK && download_left) rc = d2h_results(b, out);
        }
        const std::string keep = g_err;
        free_batch(b);
        g_err = keep;
        return rc;
    }
    svt_batch* b = nullptr;
    SVT_TRY(svt_batch_create_packed(in, device, flags, &b));
    int rc = svt_batch_genotype(b, 1);
    if (rc == SVT_OK) rc = svt_batch_results(b, out, in->n_units);
    const std::string keep = g_err;
    svt_batch_destroy(b);
    g_err = keep;
    return rc;
}

int svt_genotype_packed(const svt_packed_evidence* in, svt_result* out, int device, unsigned flags)
{
    return guarded([&] { return svt_genotype_packed_impl(in, out, device, flags); });
}

// svt_genotype_packed_from_records: canonical records in host memory -> result records, through packed evidence, with the
// host encoder running AHEAD of the wire: the batch is encoded in ranges of whole units and every finished range goes up
// (slots, slot offsets, unit headers: page-locked, straight DMA), is genotyped by its own launch of svt_packed_kernel and
// comes down while the encoder's threads are already on the next range.  The bytes are those of svt_pack_evidence +
// svt_genotype_packed; the wall time is the longer of encoding and transfer instead of their sum.
// (The producer's side of svtyper/singlesample.py:355: `sam_fragments` handed over, tallies back.)
static int svt_genotype_packed_from_records_impl(const svt_evidence_batch* in, svt_result* out, int device, unsigned flags)
{
    if (!in || (!out && in->n_units)) return fail(SVT_ERR_INVALID, "null argument");
    if (flags & ~(SVT_FLAG_SSO_ASSOCIATION | SVT_FLAG_RESULT96)) return fail(SVT_ERR_INVALID, "packed evidence takes SVT_FLAG_SSO_ASSOCIATION and SVT_FLAG_RESULT96 only");
    const uint64_t n = in->n_units;
    const bool overlap = n >= kPipelineMinUnits && n < 0x55555550ull && in->n_libs >= 1 && in->n_libs <= 256 && in->libs && in->rec_offset && in->units && in->records &&
                         in->rec_offset[0] == 0 && in->split_weight >= 0.0 && in->disc_weight >= 0.0 && std::isfinite(in->split_weight) &&
                         std::isfinite(in->disc_weight) && !std::getenv("SVT_PACKED_SERIAL");
    auto serial = [&]() -> int {   // small batches, and whatever the overlapped form declines: encode, then the packed one shot
        svt_packed_evidence* p = nullptr;
        SVT_TRY(pack_evidence(in, &p));
        const int rc = svt_genotype_packed(p, out, device, flags);
        const std::string keep = g_err;
        svt_packed_free(p);
        g_err = keep;
        return rc;
    };
    if (!overlap) return serial();
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    // the most records of any unit (the log10 table's bound) -- the encoder itself checks the offsets' monotony
    uint64_t max_f = 0;
    {
        const uint64_t kChunk = 65536, n_chunks = (n + kChunk - 1) / kChunk;
        std::vector<uint64_t> part(n_chunks, 0);
        parallel_for(n_chunks, [&](uint64_t ch) {
            uint64_t m = 0;
            for (uint64_t u = ch * kChunk; u < std::min(n, (ch + 1) * kChunk); ++u)
                if (in->rec_offset[u + 1] >= in->rec_offset[u]) m = std::max(m, in->rec_offset[u + 1] - in->rec_offset[u]);
            part[ch] = m;
        });
        for (uint64_t m : part) max_f = std::max(max_f, m);
    }
    if (max_f > 0x3FFFFFFFull) return fail(SVT_ERR_INVALID, "unit with too many records");
    const uint64_t n_rec = in->rec_offset[n];
    // 5 bytes per record (3.1 is typical; several libraries: 6, a switch in front of most pair entries of a sample sequenced more
    // than once) + a slot per stream and unit
    const uint64_t slots_cap = n_rec / 16 * (in->n_libs > 1 ? 6 : 5) + 3 * n + 4096;
    if (slots_cap >= 0xFFFFFFF0ull) return serial();

    svt_batch* b = new (std::nothrow) svt_batch();
    if (!b) return fail(SVT_ERR_NOMEM, "out of host memory");
    b->device = device;
    b->flags = flags;
    b->layout = kLayoutPacked;
    b->n_units = n;
    b->n_records = n_rec;
    svt_packed_evidence shell{};
    shell.n_units = n;
    shell.n_slots = slots_cap;
    shell.n_records = n_rec;
    shell.n_libs = in->n_libs;
    shell.libs = in->libs;
    shell.split_weight = in->split_weight;
    shell.disc_weight = in->disc_weight;
    int rc = create_packed(&shell, b, /*defer_slots=*/true, /*defer_all=*/true, max_f);

    struct Piece { uint64_t s0, s1; hipEvent_t down; };
    struct Ctx {
        svt_batch* b;
        svt_result* out;
        PipeStreams ps;
        bool out_pinned = false, r96 = false;
        void* scratch = nullptr;
        uint64_t next_slot = 0, slot_cap = 0;
        std::vector<Piece> pieces;
        ~Ctx() { g_pinned.put(scratch); }
    } ctx;
    ctx.b = b;
    ctx.out = out;
    PackedArrays arr;
    bool overflow = false;
    PackSink sink;
    // about 64 ranges: the encoder's workers never wait for one another (svt_pack.cpp, the streamed form), so small ranges only
    // cost the calling thread a hand-over each (four DMA enqueues and a launch) and leave little of the transfer exposed at the end
    // (measured, 1 M units: ranges of 250 k / 125 k / 63 k / 31 k / 16 k units -> 15.4 / 15.7 / 15.6 / 14.8 / 14.0 ms median beside
    // 17.4 for the plain sequence; with the meeting-based encoder 15.4 / 14.5 / 16.7 / 17.9 / 22.2: profiles/r04_packed_ranges.txt)
    sink.range_units = std::max<uint64_t>(8192, (n + 63) / 64);
    if (const char* e = std::getenv("SVT_PACK_RANGE_UNITS")) sink.range_units = std::max<uint64_t>(256, std::strtoull(e, nullptr, 10));
    sink.range_units = (sink.range_units + 255) / 256 * 256;      // (the encoder's chunks)
    if (rc == SVT_OK) rc = g_handles.get_stream(&ctx.ps.compute);
    if (rc == SVT_OK) rc = g_handles.get_stream(&ctx.ps.down);
    if (rc == SVT_OK) {
        ctx.r96 = (flags & SVT_FLAG_RESULT96) != 0;
        ctx.out_pinned = !ctx.r96 && g_pinned.is_pinned(out, n * sizeof(svt_result));
        if (ctx.r96) {   // tagged records: every launch writes whole workgroups' worth of slots
            ctx.slot_cap = n + ((n + sink.range_units - 1) / sink.range_units + 1) * kBlock;
            rc = ensure_result_slots(b, ctx.slot_cap);
            if (rc == SVT_OK) {
                ctx.scratch = g_pinned.get(ctx.slot_cap * sizeof(svt_result96));
                if (!ctx.scratch) rc = fail(SVT_ERR_NOMEM, "page-locked scratch for the result records");
            }
        }
    }
    if (rc == SVT_OK) {
        sink.slots_cap = slots_cap;
        sink.ctx = &ctx;
        sink.ready = [](void* vctx, const PackedArrays* a, uint64_t u0, uint64_t u1, uint64_t s0, uint64_t s1) -> int {
            Ctx& c = *static_cast<Ctx*>(vctx);
            svt_batch* b = c.b;
            if (u1 <= u0) return SVT_OK;
            b->pargs.common_mq = a->common;
            if (s1 > s0)
                HIP_TRY(hipMemcpyAsync(static_cast<char*>(b->d_records) + s0 * 16, static_cast<const char*>(a->slots) + s0 * 16, (s1 - s0) * 16,
                                       hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemcpyAsync(b->d_soff + 3 * u0, a->off + 3 * u0, (3 * (u1 - u0) + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemcpyAsync(b->d_units + u0, a->units + u0, (u1 - u0) * sizeof(svt_unit), hipMemcpyHostToDevice, b->stream));
            hipEvent_t landed, done, down;
            SVT_TRY(c.ps.event(&landed));
            HIP_TRY(hipEventRecord(landed, b->stream));
            HIP_TRY(hipStreamWaitEvent(c.ps.compute, landed, 0));
            const uint64_t r0 = c.next_slot, r1 = r0 + (c.r96 ? slots_of_launch(b, u1 - u0) : 0);
            if (c.r96 && r1 > c.slot_cap) return fail(SVT_ERR_INTERNAL, "result slots of the ranges exceed their bound");
            c.next_slot = r1;
            SVT_TRY(launch_range(b, u0, u1, c.ps.compute, r0));
            if (c.out_pinned || c.r96) {
                SVT_TRY(c.ps.event(&done));
                HIP_TRY(hipEventRecord(done, c.ps.compute));
                HIP_TRY(hipStreamWaitEvent(c.ps.down, done, 0));
                if (c.out_pinned)
                    HIP_TRY(hipMemcpyAsync(c.out + u0, b->out_dev + u0, (u1 - u0) * sizeof(svt_result), hipMemcpyDeviceToHost, c.ps.down));
                else {
                    HIP_TRY(hipMemcpyAsync(static_cast<unsigned char*>(c.scratch) + r0 * sizeof(svt_result96),
                                           reinterpret_cast<const unsigned char*>(b->out_dev) + r0 * sizeof(svt_result96),
                                           (r1 - r0) * sizeof(svt_result96), hipMemcpyDeviceToHost, c.ps.down));
                    SVT_TRY(c.ps.event(&down));
                    HIP_TRY(hipEventRecord(down, c.ps.down));
                    c.pieces.push_back(Piece{r0, r1, down});
                }
            }
            return SVT_OK;
        };
        sink.drain = [](void* vctx) {
            Ctx& c = *static_cast<Ctx*>(vctx);
            if (c.b->stream) (void)hipStreamSynchronize(c.b->stream);
            if (c.ps.compute) (void)hipStreamSynchronize(c.ps.compute);
            if (c.ps.down) (void)hipStreamSynchronize(c.ps.down);
        };
        const PackAlloc pool{[](uint64_t bytes) { return g_pinned.get(bytes); }, [](void* p) { g_pinned.put(p); }};
        rc = encode_packed(in, pool, &arr, &sink);
        overflow = rc == SVT_ERR_PACK_OVERFLOW;
    }
    // whatever was enqueued has to be through before anything is released
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (ctx.ps.compute) (void)hipStreamSynchronize(ctx.ps.compute);
    if (ctx.ps.down) (void)hipStreamSynchronize(ctx.ps.down);
    if (rc == SVT_OK) {
        b->n_slots = arr.n_slots;
        b->have_results = true;
        b->out_slots = ctx.r96 ? ctx.next_slot : n;
        Placed placed(n);
        for (const Piece& pc : ctx.pieces) expand96(static_cast<const svt_result96*>(ctx.scratch) + pc.s0, pc.s1 - pc.s0, out, placed);
        if (ctx.r96 && !placed.covers(n)) rc = fail(SVT_ERR_INTERNAL, "the device result records do not cover every unit exactly once");
        if (!ctx.out_pinned && !ctx.r96) rc = d2h_results(b, out);
    }
    g_pinned.put(arr.off);
    g_pinned.put(arr.units);
    g_pinned.put(arr.slots);
    const std::string keep = g_err;
    free_batch(b);
    g_err = keep;
    if (overflow) return serial();   // (more slots than estimated: the plain route sizes the array exactly)
    return rc;
}

int svt_genotype_packed_from_records(const svt_evidence_batch* in, svt_result* out, int device, unsigned flags)
{
    return guarded([&] { return svt_genotype_packed_from_records_impl(in, out, device, flags); });
}

void* svt_batch_stream(svt_batch* b) { return b ? (void*)b->stream : nullptr; }

void svt_batch_destroy(svt_batch* b) { free_batch(b); }

// ---- measurement hooks (tools/placement_sweep.py; not part of include/svtyper_hip.h): where a batch's records and result
// records lie in HBM decides a few per cent of the pass; these let a tool place them itself.
// `chunk_bytes` = 0: one hipMalloc; else one virtual range over physical chunks of that size (hipMemCreate / hipMemMap).
extern "C" int svt_debug_device_alloc(int device, uint64_t bytes, uint64_t chunk_bytes, void** out)
{
    return guarded([&]() -> int {
        if (!out) return fail(SVT_ERR_INVALID, "null argument");
        HIP_TRY(hipSetDevice(device));
        if (chunk_bytes) {
            uint64_t cap = 0;
            if (!g_pool.alloc_chunked(device, bytes, out, &cap, chunk_bytes)) return fail(SVT_ERR_HIP, "virtual-memory allocation failed");
            return SVT_OK;
        }
        HIP_TRY(hipMalloc(out, bytes));
        return SVT_OK;
    });
}

extern "C" int svt_debug_device_free(int device, void* p)
{
    return guarded([&]() -> int {
        if (p) g_pool.release(p, device);
        return SVT_OK;
    });
}

extern "C" int svt_debug_copy_to_host(int device, void* host, const void* dev, uint64_t bytes)
{
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
        return SVT_OK;
    });
}

extern "C" int svt_debug_memset(int device, void* p, int value, uint64_t bytes)
{
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemset(p, value, bytes));
        HIP_TRY(hipDeviceSynchronize());
        return SVT_OK;
    });
}

// the batch streams its records from `dev` (they are copied there; the buffer must hold svt_debug_record_bytes(b) bytes, 128-byte aligned)
extern "C" uint64_t svt_debug_record_bytes(const svt_batch* b)
{
    return b && b->layout == kLayoutStream ? ((uint64_t)b->sargs.last_blk + 1) * 128 : 0;
}

extern "C" void* svt_debug_records_ptr(const svt_batch* b) { return b ? const_cast<void*>(static_cast<const void*>(b->sargs.records)) : nullptr; }

extern "C" int svt_debug_bind_records(svt_batch* b, void* dev)
{
    return guarded([&]() -> int {
        if (!b || b->layout != kLayoutStream) return fail(SVT_ERR_INVALID, "canonical records only");
        if (dev && (reinterpret_cast<uintptr_t>(dev) & 127u)) return fail(SVT_ERR_INVALID, "record buffer must be 128-byte aligned");
        HIP_TRY(hipSetDevice(b->device));
        HIP_TRY(hipStreamSynchronize(b->stream));
        if (dev) {
            HIP_TRY(hipMemcpyAsync(dev, b->d_records, svt_debug_record_bytes(b), hipMemcpyDeviceToDevice, b->stream));
            HIP_TRY(hipStreamSynchronize(b->stream));
            // (the copy is checked at both ends: a mapping that silently did not take would otherwise look like bad records)
            const uint64_t total = svt_debug_record_bytes(b), probe = std::min<uint64_t>(total, 4096);
            std::vector<unsigned char> x(probe), y(probe);
            for (uint64_t at : {uint64_t(0), total - probe}) {
                HIP_TRY(hipMemcpy(x.data(), static_cast<const char*>(b->d_records) + at, probe, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(y.data(), static_cast<const char*>(dev) + at, probe, hipMemcpyDeviceToHost));
                if (std::memcmp(x.data(), y.data(), probe) != 0) return fail(SVT_ERR_HIP, "svt_debug_bind_records: the copy did not arrive");
            }
        }
        HIP_TRY(hipStreamSynchronize(b->stream));
        b->sargs.records = static_cast<const uint4*>(dev ? dev : b->d_records);
        b->have_results = false;
        return SVT_OK;
    });
}

void svt_reads_trim();      // svt_reads.cpp: the reader's pooled gather buffers

void svt_trim(void)
{
    g_pool.trim();
    g_pinned.trim();
    g_handles.trim();
    pack_trim();
    svt_reads_trim();
}

static int svt_chunk_bounds_impl(const uint64_t* rec_offset, uint64_t n_units, uint32_t group, uint64_t max_records, uint64_t* bounds,
                                 uint32_t max_chunks, uint32_t* n_chunks);

static int svt_genotype_impl(const svt_evidence_batch* in, svt_result* out, int device, unsigned flags)
{
    // a batch beyond the 32-bit record index of one resident batch (68 GB of records): chunk after chunk of whole units
    // (svt_chunk_bounds), each through this very entry point -- units are independent, so out[] is what one batch would give
    if (in && out && in->rec_offset && in->units && in->n_units &&
        (in->rec_offset[in->n_units] - in->rec_offset[0] > max_batch_records() || in->n_units > max_batch_records())) {
        for (uint64_t u = 0; u < in->n_units; ++u)
            if (in->rec_offset[u + 1] < in->rec_offset[u]) return fail(SVT_ERR_INVALID, "rec_offset not monotone");
        uint32_t n_chunks = 0;
        SVT_TRY(svt_chunk_bounds_impl(in->rec_offset, in->n_units, 1, 0, nullptr, 0, &n_chunks));
        std::vector<uint64_t> bounds((size_t)n_chunks + 1);
        SVT_TRY(svt_chunk_bounds_impl(in->rec_offset, in->n_units, 1, 0, bounds.data(), n_chunks, &n_chunks));
        std::vector<uint64_t> off;
        for (uint32_t c = 0; c < n_chunks; ++c) {
            const uint64_t lo = bounds[c], hi = bounds[c + 1], r0 = in->rec_offset[lo];
            off.resize(hi - lo + 1);
            for (uint64_t u = lo; u <= hi; ++u) off[u - lo] = in->rec_offset[u] - r0;
            svt_evidence_batch part = *in;
            part.n_units = hi - lo;
            part.rec_offset = off.data();
            part.units = in->units + lo;
            part.records = in->records ? in->records + r0 : nullptr;
            SVT_TRY(svt_genotype_impl(&part, out + lo, device, flags));
        }
        return SVT_OK;
    }
    // the streamed layout from host records: upload, pass and download overlap by unit ranges
    if (in && out && !(flags & ~kKnownFlags) && in->n_units >= kPipelineMinUnits &&
        in->n_units < 0xFFFFFFF0ull && in->rec_offset && in->units && in->records && in->n_libs >= 1 && in->n_libs <= 256 && in->libs &&
        in->rec_offset[0] == 0 && in->split_weight >= 0.0 && in->disc_weight >= 0.0 && std::isfinite(in->split_weight) &&
        std::isfinite(in->disc_weight)) {
        const int ndev = svt_device_count();
        if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
        if (device < 0 || device >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
        HIP_TRY(hipSetDevice(device));
        svt_batch* b = new (std::nothrow) svt_batch();
        if (!b) return fail(SVT_ERR_NOMEM, "out of host memory");
        b->device = device;
        b->flags = flags;
        b->layout = kLayoutStream;
        b->n_units = in->n_units;
        b->n_records = in->rec_offset[in->n_units];
        int rc = create_stream(in, b, nullptr, 0, /*defer_records=*/true);
        if (rc == SVT_OK && (b->mode == kMultiLds || b->records_resident)) {
            // library windows: the launch walks window chunks, not unit ranges -- upload in one piece (a batch without window
            // hints was uploaded by create_stream, which read the windows off the records), one launch
            if (!b->records_resident) rc = h2d_staged(b->d_records, in->records, b->n_records * sizeof(uint4), b->stream);
            if (rc == SVT_OK) rc = svt_batch_genotype(b, 1);
            if (rc == SVT_OK) rc = svt_batch_results(b, out, in->n_units);
        } else if (rc == SVT_OK) {
            bool download_left = false;
            {
            Stager st(b->stream);   // (holds this device's staging ring)
            const bool pinned = g_pinned.is_pinned(in->records, b->n_records * sizeof(uint4));
            rc = run_pipelined(b, out, &download_left, [&](uint64_t u) { return in->rec_offset[u]; },
                               [&](uint64_t i0, uint64_t i1) -> int {
                                   char* dst = static_cast<char*>(b->d_records) + i0 * 16;
                                   const char* src = reinterpret_cast<const char*>(in->records) + i0 * 16;
                                   if (pinned) { HIP_TRY(hipMemcpyAsync(dst, src, (i1 - i0) * 16, hipMemcpyHostToDevice, b->stream)); return SVT_OK; }
                                   return st.copy(dst, src, (i1 - i0) * 16);
                               });
            }
            if (rc == SVT_OK && download_left) rc = d2h_results(b, out);
        }
        const std::string keep = g_err;
        StageTimer tm;
        free_batch(b);
        tm.mark("one shot: batch released");
        g_err = keep;
        return rc;
    }
    svt_batch* b = nullptr;
    SVT_TRY(svt_batch_create(in, device, flags, &b));
    int rc = svt_batch_genotype(b, 1);
    if (rc == SVT_OK) rc = svt_batch_results(b, out, in->n_units);
    const std::string keep = g_err;
    svt_batch_destroy(b);
    g_err = keep;
    return rc;
}

int svt_genotype(const svt_evidence_batch* in, svt_result* out, int device, unsigned flags)
{
    return guarded([&] { return svt_genotype_impl(in, out, device, flags); });
}

// contiguous shards balanced by the bytes a unit costs (16 F + 112), cut only at multiples of `group` units
// (svtyper_amd/distributed.py: shard_bounds is the same rule, and tests/test_multi_device.py checks they agree)
static int svt_shard_bounds_impl(const uint64_t* rec_offset, uint64_t n_units, int n_shards, uint32_t group, uint64_t* bounds)
{
    if (!bounds || n_shards <= 0 || (n_units && !rec_offset)) return fail(SVT_ERR_INVALID, "bad arguments");
    if (group == 0) group = 1;
    bounds[0] = 0;
    const double total = n_units ? (double)(rec_offset[n_units] - rec_offset[0]) * 16.0 + 112.0 * (double)n_units : 0.0;
    for (int r = 1; r < n_shards; ++r) {
        const double target = total * (double)r / (double)n_shards;
        // first k with cost(units [0, k)) >= target
        uint64_t lo = 0, hi = n_units;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            const double c = (double)(rec_offset[mid] - rec_offset[0]) * 16.0 + 112.0 * (double)mid;
            if (c < target) lo = mid + 1; else hi = mid;
        }
        uint64_t k = lo / group * group;
        k = std::min<uint64_t>(n_units, std::max<uint64_t>(bounds[r - 1], k));
        bounds[r] = k;
    }
    bounds[n_shards] = n_units;
    return SVT_OK;
}

int svt_shard_bounds(const uint64_t* rec_offset, uint64_t n_units, int n_shards, uint32_t group, uint64_t* bounds)
{
    return guarded([&] { return svt_shard_bounds_impl(rec_offset, n_units, n_shards, group, bounds); });
}

// greedy cut into the fewest chunks of at most `max_records` records (and units), at multiples of `group` units
static int svt_chunk_bounds_impl(const uint64_t* rec_offset, uint64_t n_units, uint32_t group, uint64_t max_records, uint64_t* bounds,
                                 uint32_t max_chunks, uint32_t* n_chunks)
{
    if (!n_chunks || (n_units && !rec_offset)) return fail(SVT_ERR_INVALID, "null argument");
    if (group == 0) group = 1;
    if (max_records == 0 || max_records > max_batch_records()) max_records = max_batch_records();
    uint32_t count = 0;
    uint64_t lo = 0;
    if (bounds && max_chunks) bounds[0] = 0;
    while (lo < n_units) {
        // the last k <= n_units with records[lo, k) <= max_records and k - lo <= max_records: rec_offset is monotone
        uint64_t a = lo, b = std::min<uint64_t>(n_units, lo + max_records);
        while (a < b) {
            const uint64_t mid = a + (b - a + 1) / 2;
            if (rec_offset[mid] - rec_offset[lo] <= max_records) a = mid; else b = mid - 1;
        }
        uint64_t hi = a == n_units ? n_units : lo + (a - lo) / group * group;
        if (hi <= lo) return fail(SVT_ERR_INVALID, "svt_chunk_bounds: the units of one site hold more records than a batch can");
        ++count;
        if (bounds) {
            if (count > max_chunks) return fail(SVT_ERR_INVALID, "svt_chunk_bounds: bounds[] is too short");
            bounds[count] = hi;
        }
        lo = hi;
    }
    *n_chunks = count;
    return SVT_OK;
}

int svt_chunk_bounds(const uint64_t* rec_offset, uint64_t n_units, uint32_t group, uint64_t max_records, uint64_t* bounds,
                     uint32_t max_chunks, uint32_t* n_chunks)
{
    return guarded([&] { return svt_chunk_bounds_impl(rec_offset, n_units, group, max_records, bounds, max_chunks, n_chunks); });
}

static int svt_genotype_multi_impl(const svt_evidence_batch* in, svt_result* out, const int* devices, int n_devices,
                                   uint32_t group, unsigned flags)
{
    if (!in || !devices || n_devices <= 0 || n_devices > 64) return fail(SVT_ERR_INVALID, "bad device list");
    const uint64_t n = in->n_units;
    if (n && (!out || !in->rec_offset || !in->units)) return fail(SVT_ERR_INVALID, "null argument");
    const int ndev = svt_device_count();
    if (ndev <= 0) return fail(SVT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    for (int d = 0; d < n_devices; ++d)
        if (devices[d] < 0 || devices[d] >= ndev) return fail(SVT_ERR_NO_DEVICE, "device index out of range");
    for (uint64_t u = 0; u < n; ++u)
        if (in->rec_offset[u + 1] < in->rec_offset[u]) return fail(SVT_ERR_INVALID, "rec_offset not monotone");
    std::vector<uint64_t> bounds((size_t)n_devices + 1);
    SVT_TRY(svt_shard_bounds_impl(in->rec_offset, n, n_devices, group, bounds.data()));
    std::vector<int> rc((size_t)n_devices, SVT_OK);
    std::vector<std::string> msg((size_t)n_devices);
    // one host thread per device: upload of its shard, ONE pass, download -- the threads only share the
    // caller's read-only arrays and write disjoint ranges of out[]
    run_threads((unsigned)n_devices, [&](unsigned t) {
        const uint64_t lo = bounds[t], hi = bounds[t + 1];
        if (lo == hi) return;
        std::vector<uint64_t> off(hi - lo + 1);
        const uint64_t r0 = in->rec_offset[lo];
        for (uint64_t u = lo; u <= hi; ++u) off[u - lo] = in->rec_offset[u] - r0;
        svt_evidence_batch shard = *in;
        shard.n_units = hi - lo;
        shard.rec_offset = off.data();
        shard.units = in->units + lo;
        shard.records = in->records ? in->records + r0 : nullptr;
        rc[t] = svt_genotype(&shard, out + lo, devices[t], flags);
        if (rc[t] != SVT_OK) msg[t] = g_err;     // (g_err is thread-local: hand the text to the calling thread)
    });
    for (int d = 0; d < n_devices; ++d)
        if (rc[d] != SVT_OK) return fail(rc[d], "device " + std::to_string(devices[d]) + ": " + msg[d]);
    return SVT_OK;
}

int svt_genotype_multi(const svt_evidence_batch* in, svt_result* out, const int* devices, int n_devices, uint32_t group,
                       unsigned flags)
{
    return guarded([&] { return svt_genotype_multi_impl(in, out, devices, n_devices, group, flags); });
}

}  // extern "C"
