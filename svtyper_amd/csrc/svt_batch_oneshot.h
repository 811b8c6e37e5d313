// svt_batch_oneshot.h -- part of the single translation unit svtyper_hip.hip (included there, in order; not a stand-alone header):
// the pipelined one shot (upload || pass || download by unit ranges), 96-byte record expansion, the placement audition.

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

