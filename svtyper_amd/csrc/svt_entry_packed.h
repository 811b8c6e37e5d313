// svt_entry_packed.h -- part of the single translation unit svtyper_hip.hip (included there, in order; not a stand-alone header):
// C ABI: page-locked buffers, svt_pack_evidence, packed batches, svt_genotype_packed*.

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

