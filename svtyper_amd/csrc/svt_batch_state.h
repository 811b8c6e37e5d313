// svt_batch_state.h -- part of the single translation unit svtyper_hip.hip (included there, in order; not a stand-alone header):
// the resident batch's device scratch pools, the workgroup plan (which kernel, how many units per workgroup, which small-launch kernel) and the pass launch.


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

