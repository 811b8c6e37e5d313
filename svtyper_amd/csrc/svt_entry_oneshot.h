// svt_entry_oneshot.h -- part of the single translation unit svtyper_hip.hip (included there, in order; not a stand-alone header):
// C ABI: svt_trim, svt_genotype (one shot), svt_shard_bounds / svt_chunk_bounds, svt_genotype_multi.

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
        in->n_units < 0xFFFFFFF0ull && in->rec_offset && in->units && in->records && in->n_libs >= 1 && in->n_libs <= 65536 && in->libs &&
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

