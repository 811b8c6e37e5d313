// svt_batch_packed.h -- part of the single translation unit svtyper_hip.hip (included there, in order; not a stand-alone header):
// packed evidence: the page-locked pool, svt_pack_evidence's owner object, the packed batch.

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

