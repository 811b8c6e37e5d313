// svt_batch_create.h -- part of the single translation unit svtyper_hip.hip (included there, in order; not a stand-alone header):
// svt_batch_create's work: validation of the units, host-built tables, library-window grouping (hints or svt_window_scan_kernel), the upload of the canonical CSR, the LDS / occupancy budget.

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
                if ((U.libs >> 24) != 0 || (U.flags & ~SVT_UNIT_SKIP)) p.bad |= 8;
                const uint32_t w_lo = SVT_UNIT_LIBS_FIRST(U.libs), w_cnt = SVT_UNIT_LIBS_COUNT(U.libs);
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
    const uint32_t whole_key = SVT_UNIT_LIBS(0u, in->n_libs);   // (n_libs <= 255 here)
    const bool may_window = in->n_libs > 1 && T.fast_geometry && !(b->flags & SVT_FLAG_GENERAL_TABLES);
    bool windowed = may_window && (all_hinted || whole_batch_window);
    // No hints and too many libraries for one window: the windows are read off the records themselves, on the device,
    // right after the upload (svt_window_scan_kernel.h) -- not when the caller uploads the records later (pipelined one-shot)
    const bool derive_windows = may_window && !windowed && n > 0 && T.narrow_bins;
    // (the pipelined one-shot of such a batch: its pass is a few tenths of a millisecond beside tens of milliseconds of upload, so
    // nothing is lost by uploading first and reading the windows -- the general mode it used to take instead runs at a third of
    // the window kernel's speed)
    if (derive_windows) defer_records = false;
    b->records_resident = !defer_records;
    // group the units by window (a counting sort on the window's first library: stable, original order inside a group) and cut
    // the groups into chunks.  `hint_of(u)` = the unit's window as SVT_UNIT_LIBS(first, count).  Units whose windows start at the
    // same library but differ in length (scanned windows: the units of one sample need not all use its last library) share the
    // longest of them: a window is what a workgroup STAGES, and the longer one holds every library the shorter ones name.
    auto group_units = [&](auto&& hint_of) -> bool {
        std::vector<uint32_t> start(65537, 0u);
        std::vector<uint16_t> count_of(65536, 0);
        for (uint64_t u = 0; u < n; ++u) {
            const uint32_t h = hint_of(u), first = SVT_UNIT_LIBS_FIRST(h), cnt = SVT_UNIT_LIBS_COUNT(h);
            if (cnt == 0) return false;
            count_of[first] = std::max(count_of[first], (uint16_t)cnt);
            ++start[first + 1];
        }
        for (uint32_t k = 0; k < 65536u; ++k) start[k + 1] += start[k];
        perm.resize(n);
        groups.clear();
        {
            std::vector<uint32_t> at(start.begin(), start.end() - 1);
            for (uint64_t u = 0; u < n; ++u) perm[at[SVT_UNIT_LIBS_FIRST(hint_of(u))]++] = (uint32_t)u;
        }
        for (uint32_t k = 0; k < 65536u; ++k) {
            if (start[k + 1] == start[k]) continue;
            const uint32_t lo = k, cnt = count_of[k];
            WgDesc w{};
            w.lib_lo = lo;
            w.lib_cnt = cnt;
            w.bin_lo = T.libs[lo].tab_off;
            w.bin_cnt = T.libs[lo + cnt - 1].tab_off + T.libs[lo + cnt - 1].n_bins + 1 - w.bin_lo;
            max_win_bins = std::max(max_win_bins, w.bin_cnt);
            max_win_libs = std::max(max_win_libs, w.lib_cnt);
            groups.push_back(Group{start[k], start[k + 1], w});
        }
        return true;
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
        windowed = group_units([&](uint64_t u) -> uint32_t { return all_hinted ? in->units[u].libs : whole_key; });
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
        for (uint64_t u = 0; u < n && all_seen; ++u) all_seen = seen[u] != 0u && SVT_UNIT_LIBS_FIRST(seen[u]) + SVT_UNIT_LIBS_COUNT(seen[u]) <= in->n_libs;
        if (all_seen) {
            windowed = group_units([&](uint64_t u) -> uint32_t { return seen[u]; });
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
    // the general mode keeps one 32-byte descriptor per library in LDS: beyond 1 024 libraries a batch has to come with library
    // windows (svt_unit.libs, one per sample -- what every producer in this repository writes) or with windows the scan can derive
    if (b->mode == kGeneral && in->n_libs > 1024)
        return fail(SVT_ERR_UNSUPPORTED, "more than 1024 libraries in a batch without usable library windows (svt_unit.libs)");
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
    b->resident_wgs = (uint32_t)std::min<size_t>((size_t)b->wgs_per_cu, (160 * 1024) / (tables + kWavesPerBlock * kStreamRingBytes)) *
                      cu_count(b->device);
    b->out_dev = b->d_out;
    b->out_slots = !a.result96 ? n : b->mode == kMultiLds ? (uint64_t)b->n_chunks * kBlock * (uint64_t)b->window_tiles : slots_of_launch(b, n);
    SVT_TRY(ensure_result_slots(b, b->out_slots));
    a.out = b->d_out;
    a.err = b->d_err;
    a.lib0 = T.libs[0];
    fill_gt_consts(a.c, in->split_weight, in->disc_weight);
    b->out_dev = b->d_out;   // svt_batch_device_results / svt_batch_bind_device_results / svt_batch_site_qual
    b->lds_bytes = tables + kWavesPerBlock * kStreamRingBytes;
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

