// svt_reads.cpp -- native BAM access + fragment summariser (include/svtyper_reads.h).
//
// Host-only C++ (no HIP): BGZF/BAM/BAI reader with the fetch()/count() semantics the SVTyper path
// relies on (pysam's, as restated in svtyper_amd/bam.py), read-fragment assembly and split-read QC
// (svtyper/parsers.py:729-768, 891-1058 as restated in svtyper_amd/fragments.py) and the emission of
// svt_fragment summaries (svtyper_amd/geometry.py).  The Python modules are the portable
// implementation and the checker of this file (tests/test_native_reads.py compares the summaries
// byte for byte); this file exists because per-read Python objects, not the GPU, bound a real run.

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/svtyper_reads.h"
#include "svt_error.h"
#include "svt_geometry_math.h"
#include "svt_host_cpus.h"

namespace {

using svt::fail;
using svt::guarded;
using svt::run_threads;

std::atomic<double> g_cpu_s_per_unit{0.0};   // CPU seconds per unit of the last svt_bam_summarise / svt_bam_evidence call on any file

// ------------------------------------------------------------------------------------------
// BGZF: random access through (compressed offset << 16 | in-block offset) addresses
// ------------------------------------------------------------------------------------------
// the whole file, mapped read-only once per handle and shared by all worker threads: no read()
// syscalls or stdio buffers on the fetch path, the inflate input is the mapping itself
struct FileMap {
    const uint8_t* data = nullptr;
    size_t size = 0;
    FileMap() = default;
    FileMap(const FileMap&) = delete;
    FileMap& operator=(const FileMap&) = delete;
    ~FileMap() { if (data) munmap(const_cast<uint8_t*>(data), size); }
    bool open(const std::string& path)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size <= 0) { ::close(fd); return false; }
        void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) return false;
        madvise(p, (size_t)st.st_size, MADV_RANDOM);   // region fetches, not a scan
        data = static_cast<const uint8_t*>(p);
        size = (size_t)st.st_size;
        return true;
    }
};

// Raw-deflate decoding of BGZF blocks is where a region fetch spends its time on real data (a window is reached by
// inflating every block from the start of its 16-kb bin).  libdeflate's whole-buffer decoder is 2-3x faster
// than zlib's streaming one; the image ships its runtime (libdeflate.so.0) without headers, so it is bound by
// name at first use and zlib stays as the decoder when it is absent (or SVT_INFLATE=zlib asks for it).
struct FastInflate {
    void* (*alloc)() = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    FastInflate()
    {
        const char* want = std::getenv("SVT_INFLATE");
        if (want && std::strcmp(want, "zlib") == 0) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
        decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(
            dlsym(h, "libdeflate_deflate_decompress"));
        release = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
        if (!alloc || !decompress || !release) alloc = nullptr;
    }
    bool usable() const { return alloc != nullptr; }
};
static const FastInflate& fast_inflate()
{
    static const FastInflate f;
    return f;
}

// One inflated BGZF block: immutable once it is published, so readers on several threads can hold it.
struct BlockData {
    std::vector<uint8_t> data;
    uint64_t next = 0;         // compressed offset of the block behind it (== its own offset: end of data / unusable block)
};
typedef std::shared_ptr<const BlockData> BlockRef;

// Inflated blocks shared by the worker threads of ONE svt_bam_summarise call.  Workers take runs of neighbouring units,
// and a worker that starts a run walks up to its first window through the blocks in front of it (a window is reached from
// the start of its 16-kb bin: seven 64-KiB blocks at 30x on average) -- blocks the worker of the run before inflates too,
// for its own last units.  Measured on 290 whole-genome-like sites: 24 % (8 workers) to 54 % (15) of all inflate calls were
// such repeats, and inflate is three quarters of the reader's time there.  A block is looked up here after the reader's own
// slots missed and published after it was inflated: one short critical section per 64 KiB of records.  64 shards x 16 ways
// = 1 024 blocks (64 MiB) at most, first-in-first-out per shard; a block a reader still holds outlives its eviction.
class SharedBlocks {
public:
    // The block at `coff`, or null with *claimed = true: the caller inflates it and then calls publish() or abandon().
    // While one worker inflates a block the others that want it wait here instead of inflating it too (workers start
    // their runs side by side: with 47 of them 2 291 inflate calls for 929 blocks before this).
    // With `in_flight` the call does not wait: a block somebody else is inflating comes back as null, *in_flight = true.
    BlockRef find_or_claim(uint64_t coff, bool* claimed, bool* in_flight = nullptr)
    {
        Shard& sh = shard(coff);
        std::unique_lock<std::mutex> g(sh.lock);
        *claimed = false;
        if (in_flight) *in_flight = false;
        for (;;) {
            int at = -1;
            for (int i = 0; i < kWays; ++i)
                if (sh.coff[i] == coff) { at = i; break; }
            if (at >= 0 && sh.block[at]) return sh.block[at];
            if (at < 0) {                                    // nobody has it, nobody is on it: the caller's
                sh.coff[sh.clock] = coff;
                sh.block[sh.clock].reset();
                sh.clock = (sh.clock + 1) % kWays;
                *claimed = true;
                return BlockRef();
            }
            if (in_flight) {
                *in_flight = true;
                return BlockRef();
            }
            sh.ready.wait(g);                                // in flight: published, abandoned or pushed out when we wake
        }
    }
    void publish(uint64_t coff, const BlockRef& b)
    {
        Shard& sh = shard(coff);
        {
            std::lock_guard<std::mutex> g(sh.lock);
            int at = -1;
            for (int i = 0; i < kWays; ++i)
                if (sh.coff[i] == coff) { at = i; break; }
            if (at < 0) {                                    // (its place went to sixteen newer blocks meanwhile)
                at = sh.clock;
                sh.clock = (sh.clock + 1) % kWays;
                sh.coff[at] = coff;
            }
            sh.block[at] = b;
        }
        sh.ready.notify_all();
    }
    void abandon(uint64_t coff)                              // the block is unusable: whoever waits finds that out for itself
    {
        Shard& sh = shard(coff);
        {
            std::lock_guard<std::mutex> g(sh.lock);
            for (int i = 0; i < kWays; ++i)
                if (sh.coff[i] == coff && !sh.block[i]) sh.coff[i] = ~0ull;
        }
        sh.ready.notify_all();
    }

private:
    static constexpr int kShards = 64, kWays = 16;
    struct Shard {
        std::mutex lock;
        std::condition_variable ready;
        uint64_t coff[kWays];
        BlockRef block[kWays];                               // null under a valid offset: being inflated
        int clock = 0;
        Shard() { for (auto& c : coff) c = ~0ull; }
    };
    Shard& shard(uint64_t coff) { return shards_[(coff * 0x9E3779B97F4A7C15ull) >> 58]; }
    Shard shards_[kShards];
};

class Bgzf {
public:
    explicit Bgzf(const FileMap& file, SharedBlocks* shared = nullptr) : file_(file), shared_(shared), empty_(std::make_shared<BlockData>())
    {
        std::memset(&zs_, 0, sizeof zs_);
        if (fast_inflate().usable()) fast_ = fast_inflate().alloc();
        if (!fast_) zs_ok_ = inflateInit2(&zs_, -15) == Z_OK;   // one inflate state per reader, reset per block
        block_ = empty_.get();
    }
    ~Bgzf()
    {
        if (fast_) fast_inflate().release(fast_);
        if (zs_ok_) inflateEnd(&zs_);
    }
    Bgzf(const Bgzf&) = delete;
    Bgzf& operator=(const Bgzf&) = delete;
    bool ok() const { return file_.data != nullptr && (zs_ok_ || fast_); }
    bool failed() const { return bad_; }
    void mark_bad() { bad_ = true; }   // the record stream inside the blocks is corrupt
    uint64_t n_inflated = 0, n_shared_hits = 0, n_ahead = 0;   // (SVT_TRACE)
    double inflate_s = 0.0;

    void seek(uint64_t voff)
    {
        load(voff >> 16);
        uoff_ = (size_t)(voff & 0xFFFF);
    }
    uint64_t tell() const
    {
        if (uoff_ >= block_->data.size() && !block_->data.empty()) return block_->next << 16;
        return (coff_ << 16) | uoff_;
    }
    // returns the number of bytes actually read
    size_t read(void* dst, size_t n)
    {
        size_t got = 0;
        uint8_t* out = static_cast<uint8_t*>(dst);
        while (got < n) {
            const size_t avail = block_->data.size() - std::min(uoff_, block_->data.size());
            if (avail == 0) {
                const uint64_t next = block_->next;
                if (block_ != empty_.get() && next == coff_) break;
                if (!load(next)) break;
                uoff_ = 0;
                continue;
            }
            const size_t take = std::min(avail, n - got);
            std::memcpy(out + got, block_->data.data() + uoff_, take);
            uoff_ += take;
            got += take;
        }
        return got;
    }

    // n bytes at the read position as one span inside the current inflated block, or nullptr when they
    // straddle a block boundary / the file ends (the caller then falls back to read()).  The pointer stays
    // valid until the next call that may load a block.
    const uint8_t* contiguous(size_t n)
    {
        if (uoff_ >= block_->data.size()) {
            const uint64_t next = block_->next;
            if (block_ != empty_.get() && next == coff_) return nullptr;
            if (!load(next)) return nullptr;
            uoff_ = 0;
        }
        return block_->data.size() - uoff_ >= n ? block_->data.data() + uoff_ : nullptr;
    }
    void advance(size_t n) { uoff_ += n; }   // over bytes contiguous() has just vouched for

private:
    // Recently used blocks of this reader: the two windows of a unit and its neighbours walk forward through the same
    // blocks, and a list of sites often comes back to a region (both ends of a large event, overlapping calls, the same
    // targets again).  kSlots references, 8 MiB of blocks at most; the offsets sit in an array of their own: a look-up is
    // one pass over 1 KiB.  (32 slots: a cycle over ~50 blocks -- the fixture's 211 sites, repeated -- missed on every
    // third site, a third of the reader's time.)
    static constexpr int kSlots = 128;
    int slot_for(uint64_t coff) const
    {
        for (int i = 0; i < n_slots_; ++i)
            if (coffs_[i] == coff) return i;
        return -1;
    }
    bool use(uint64_t coff, const BlockRef& b)           // make `b` the current block, remembered under `coff`
    {
        int i;
        if (n_slots_ < kSlots) i = n_slots_++;
        else {
            i = clock_;
            clock_ = (clock_ + 1) % kSlots;
        }
        coffs_[i] = coff;
        slots_[i] = b;
        block_ = b.get();
        coff_ = coff;
        return !block_->data.empty() || block_->next > coff;
    }
    bool park(uint64_t coff, bool is_bad)                // end of file / unusable block: an empty block that is its own successor
    {
        if (is_bad) bad_ = true;
        auto b = std::make_shared<BlockData>();
        b->next = coff;
        use(coff, b);
        return false;
    }
    // The block at `coff` inflated into a fresh BlockData; null with *unusable = false at the end of the file, null with
    // *unusable = true for a header that cannot be one.  A stream that does not inflate still returns its block (the
    // bytes stay readable) with *unusable = true.
    std::shared_ptr<BlockData> inflate_block(uint64_t coff, bool* unusable)
    {
        *unusable = false;
        if (coff + 18 > file_.size) return nullptr;                    // end of file
        *unusable = true;
        const uint8_t* hdr = file_.data + coff;
        if (hdr[0] != 31 || hdr[1] != 139) return nullptr;
        const size_t xlen = hdr[10] | (hdr[11] << 8);
        if (coff + 12 + xlen > file_.size) return nullptr;
        int bsize = -1;
        for (size_t i = 0; i + 4 <= xlen;) {
            const uint8_t* x = hdr + 12 + i;
            const size_t slen = x[2] | (x[3] << 8);
            if (x[0] == 66 && x[1] == 67 && i + 6 <= xlen) bsize = x[4] | (x[5] << 8);
            i += 4 + slen;
        }
        if (bsize < 0 || coff + (uint64_t)bsize + 1 > file_.size) return nullptr;
        const int clen = bsize - (int)xlen - 19;
        if (clen < 0) return nullptr;
        const uint8_t* cdata = hdr + 12 + xlen;
        const uint8_t* tail = cdata + clen;
        const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (isize > 65536u) return nullptr;              // a BGZF block inflates to at most 64 KiB
        auto b = std::make_shared<BlockData>();
        b->data.resize(isize);
        b->next = coff + (uint64_t)bsize + 1;
        bool inflated = true;
        const auto t_inflate = std::chrono::steady_clock::now();
        if (isize && fast_) {
            // exactly isize bytes or an error (a null "actual size" pointer makes a short stream a failure)
            if (fast_inflate().decompress(fast_, cdata, (size_t)clen, b->data.data(), b->data.size(), nullptr) != 0) inflated = false;
        } else if (isize) {
            if (inflateReset(&zs_) != Z_OK) inflated = false;
            else {
                zs_.next_in = const_cast<Bytef*>(cdata);
                zs_.avail_in = (uInt)clen;
                zs_.next_out = b->data.data();
                zs_.avail_out = (uInt)b->data.size();
                if (inflate(&zs_, Z_FINISH) != Z_STREAM_END) inflated = false;
            }
        }
        inflate_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_inflate).count();
        ++n_inflated;
        *unusable = !inflated;
        return b;
    }
    // offset of the block behind the one at `coff`, from its header alone; 0 when there is none to be had
    uint64_t next_offset(uint64_t coff) const
    {
        if (coff + 18 > file_.size) return 0;
        const uint8_t* hdr = file_.data + coff;
        if (hdr[0] != 31 || hdr[1] != 139) return 0;
        const size_t xlen = hdr[10] | (hdr[11] << 8);
        if (coff + 12 + xlen > file_.size) return 0;
        for (size_t i = 0; i + 4 <= xlen;) {
            const uint8_t* x = hdr + 12 + i;
            if (x[0] == 66 && x[1] == 67 && i + 6 <= xlen) return coff + (uint64_t)(x[4] | (x[5] << 8)) + 1;
            i += 4 + (size_t)(x[2] | (x[3] << 8));
        }
        return 0;
    }
    struct Claim {                                       // a claimed block that is not published is given up on every way out
        SharedBlocks* shared = nullptr;
        uint64_t coff = 0;
        ~Claim() { if (shared) shared->abandon(coff); }
    };
    // Somebody else is inflating the block this reader needs next.  Readers walk forward, so the blocks behind it are
    // wanted too -- by this reader, and by the one it waits for: instead of waiting, inflate the first of the next
    // kAhead blocks nobody has or is on.  Workers that walk up to neighbouring windows through the same blocks thereby
    // inflate them side by side instead of queueing behind one another (47 workers on 290 whole-genome-like sites spent
    // two thirds of their time in that queue).  False when there was nothing to do.
    bool help_ahead(uint64_t coff)
    {
        static constexpr int kAhead = 12;
        uint64_t c = coff;
        for (int k = 0; k < kAhead; ++k) {
            c = next_offset(c);
            if (c == 0 || c + 18 > file_.size) return false;
            if (slot_for(c) >= 0) continue;
            bool claimed = false, in_flight = false;
            if (shared_->find_or_claim(c, &claimed, &in_flight) || in_flight) continue;
            Claim claim{shared_, c};
            bool unusable = false;
            std::shared_ptr<BlockData> b = inflate_block(c, &unusable);
            if (!b || unusable) return false;            // (left to the reader that gets there: it reports the failure)
            shared_->publish(c, b);
            claim.shared = nullptr;
            ++n_ahead;
            return true;
        }
        return false;
    }
    bool load(uint64_t coff)
    {
        const int hit = slot_for(coff);
        if (hit >= 0) {
            block_ = slots_[hit].get();
            coff_ = coff;
            return !block_->data.empty() || block_->next > coff;
        }
        Claim claim;
        if (shared_) {
            for (;;) {
                bool claimed = false, in_flight = false;
                if (BlockRef b = shared_->find_or_claim(coff, &claimed, &in_flight)) { ++n_shared_hits; return use(coff, b); }
                if (claimed) { claim.shared = shared_; claim.coff = coff; break; }
                if (help_ahead(coff)) continue;          // (in flight elsewhere: useful work first, then look again)
                if (BlockRef b = shared_->find_or_claim(coff, &claimed)) { ++n_shared_hits; return use(coff, b); }   // waits
                if (claimed) { claim.shared = shared_; claim.coff = coff; }
                break;
            }
        }
        bool unusable = false;
        std::shared_ptr<BlockData> b = inflate_block(coff, &unusable);
        if (!b) return park(coff, unusable);
        if (unusable) bad_ = true;                       // (its bytes stay readable, as before: the caller sees failed())
        if (claim.shared && !unusable) { shared_->publish(coff, b); claim.shared = nullptr; }
        use(coff, b);
        return true;
    }

    const FileMap& file_;
    SharedBlocks* shared_;
    z_stream zs_;
    bool zs_ok_ = false;
    void* fast_ = nullptr;   // libdeflate decompressor of this reader
    BlockRef slots_[kSlots];
    uint64_t coffs_[kSlots];
    int n_slots_ = 0;
    int clock_ = 0;
    std::shared_ptr<BlockData> empty_;
    const BlockData* block_ = nullptr;
    uint64_t coff_ = 0;
    size_t uoff_ = 0;
    bool bad_ = false;
};

// ------------------------------------------------------------------------------------------
// CIGAR helpers (svtyper_amd/fragments.py, svtyper/parsers.py:922-947,1062-1101,1242-1253)
// ------------------------------------------------------------------------------------------
typedef std::vector<std::pair<int, int64_t>> Cigar;   // (op, len)
inline bool is_clip(int op) { return op == 4 || op == 5; }
inline bool consumes_ref(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
inline bool consumes_query(int op) { return op == 0 || op == 1 || op == 7 || op == 8; }
inline bool is_aligned(int op) { return op == 0 || op == 7 || op == 8; }

struct QueryPos { int64_t start = 0, end = 0, length = 0; };

QueryPos query_pos_from_cigar(const Cigar& cigar, bool reverse)
{
    QueryPos q;
    const size_t n = cigar.size();
    for (size_t i = 0; i < n; ++i) {
        const auto& c = reverse ? cigar[n - 1 - i] : cigar[i];
        if (is_clip(c.first)) {
            if (i == 0) { q.start += c.second; q.end += c.second; }
            q.length += c.second;
        } else if (consumes_query(c.first)) {
            q.end += c.second;
            q.length += c.second;
        }
    }
    return q;
}

bool left_clipped(const Cigar& c)
{
    const bool lc = is_clip(c.front().first), rc = is_clip(c.back().first);
    return (lc && !rc) || (lc && rc && c.front().second > c.back().second);
}

bool parse_cigar_string(const char* s, size_t n, Cigar& out)
{
    static const char* ops = "MIDNSHP=X";
    int64_t num = 0;
    bool have = false;
    for (size_t i = 0; i < n; ++i) {
        const char ch = s[i];
        if (ch == 0) return false;      // (strchr would find the terminator of `ops`)
        if (ch >= '0' && ch <= '9') { num = num * 10 + (ch - '0'); have = true; continue; }
        const char* p = std::strchr(ops, ch);
        if (!p || !have) return false;
        out.emplace_back((int)(p - ops), num);
        num = 0;
        have = false;
    }
    return !have;
}

// ------------------------------------------------------------------------------------------
// reads, pieces, fragments
// ------------------------------------------------------------------------------------------
struct Piece {
    int32_t tid = 0;          // -2: dummy piece (chrom None); -3: chromosome not in the header
    int64_t start = 0, end = 0;
    bool reverse = false;
    int64_t mapq = 0;
    const Cigar* cigar = nullptr;   // not owned: the record's own operations, or split_candidate's scratch for an SA entry --
    QueryPos qp;                    // a piece is copied around (left / right) and lives only until its PieceOut is taken
};

struct ReadInfo {                // what a primary read contributes to a summary (svt_read_summary)
    int32_t tid = -1;
    int64_t start = 0, end = 0;
    bool reverse = false;
    int mapq = 0;
    int n_iv = 0;                // the (at most two) gap-free aligned intervals closest to the unit's breakends
    int64_t iv_start[2] = {0, 0}, iv_end[2] = {0, 0};
};

struct PieceOut {                // what a split piece contributes to a summary (svt_piece_summary)
    int32_t tid = 0;
    int64_t start = 0, end = 0, mapq = 0;
    bool reverse = false;
};

struct SplitOut {
    bool soft = false;
    PieceOut left, right;
};

struct Split {
    bool soft = false;
    Piece left, right;
};

struct Fragment {                // reused from unit to unit (its vectors keep their capacity)
    int lib = 0;
    int num_primary = 0;
    uint32_t name_off = 0, name_len = 0;   // query name in the workspace's name arena
    std::vector<uint16_t> seen;            // flags already added under this query name (parsers.py:748-754)
    std::vector<ReadInfo> primaries;
    std::vector<SplitOut> splits;
    void reset(int library, uint32_t off, uint32_t len)
    {
        lib = library;
        num_primary = 0;
        name_off = off;
        name_len = len;
        seen.clear();
        primaries.clear();
        splits.clear();
    }
};

struct Record {               // one BAM alignment, decoded as far as the path needs
    int32_t tid = -1;
    int64_t pos = 0, end = 0;
    uint16_t flag = 0;
    int mapq = 0;
    int64_t l_seq = 0;
    int64_t tlen = 0;         // template_length
    const char* name = "";    // query name: points into the record's bytes (the inflated block, or the caller's gather buffer for a
    uint32_t name_len = 0;    // record that straddles blocks): valid until the next record is read
    Cigar cigar;
    const uint8_t* tags = nullptr;
    size_t tags_len = 0;
    std::string name_str() const { return std::string(name, name_len); }
};

// Z-typed tag value or nullptr; walks the tag area like svtyper_amd/bam.py::_parse_tags.
// A kept read is asked for RG and then for SA: the second search need not walk the tags in front of RG again.  `from`: where
// the walk starts; `resume` (optional): set to the offset behind the found tag; `other` (optional, with o0 o1): the first Z
// value of that other tag met ON THE WAY to the found one.  First-match semantics are those of two full walks (search RG from
// 0 noting SA, then -- when SA was not met -- search SA from `resume`); the tags VALIDATED are every tag of the read either
// way: the caller walks what lies behind the second tag it found as well (`validate_only`), so a malformed tag anywhere
// fails the call with SVT_ERR_INVALID -- as svtyper_amd/bam.py::_parse_tags, which parses the whole tag area of every read it
// keeps, raises (tests/test_native_reads.py::test_truncated_tag_behind_rg_is_malformed_in_both_tag_orders).
const char* find_z_tag(const Record& r, char k0, char k1, bool* malformed, size_t from = 0, size_t* resume = nullptr,
                       char o0 = 0, char o1 = 0, const char** other = nullptr, bool validate_only = false)
{
    const uint8_t* b = r.tags;
    size_t i = from, n = r.tags_len;
    while (i + 3 <= n) {
        const char a0 = (char)b[i], a1 = (char)b[i + 1], t = (char)b[i + 2];
        i += 3;
        size_t skip = 0;
        switch (t) {
        case 'A': case 'c': case 'C': skip = 1; break;
        case 's': case 'S': skip = 2; break;
        case 'i': case 'I': case 'f': skip = 4; break;
        case 'Z': case 'H': {
            // (the values are short -- RG ids, MD strings of a few characters: a library call per tag costs more than the scan)
            const uint8_t* q = b + i;
            const uint8_t* const near_end = b + std::min(n, i + 24);
            while (q < near_end && *q) ++q;
            const void* z = (q < near_end) ? q : (q < b + n ? std::memchr(q, 0, (size_t)(b + n - q)) : nullptr);
            if (!z) { *malformed = true; return nullptr; }
            skip = (size_t)(static_cast<const uint8_t*>(z) - (b + i)) + 1;
            if (!validate_only && a0 == k0 && a1 == k1 && t == 'Z') {
                if (resume) *resume = i + skip;
                return reinterpret_cast<const char*>(b + i);
            }
            if (other && !*other && a0 == o0 && a1 == o1 && t == 'Z') *other = reinterpret_cast<const char*>(b + i);
            break;
        }
        case 'B': {
            if (i + 5 > n) { *malformed = true; return nullptr; }
            const char sub = (char)b[i];
            const uint32_t cnt = b[i + 1] | (b[i + 2] << 8) | (b[i + 3] << 16) | ((uint32_t)b[i + 4] << 24);
            const size_t sz = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            skip = 5 + (size_t)cnt * sz;
            break;
        }
        default: *malformed = true; return nullptr;
        }
        i += skip;
    }
    return nullptr;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// the BAM handle: header + index (shared, read-only); file handles are per thread
// ------------------------------------------------------------------------------------------
struct svt_bam {
    std::string path;
    FileMap file;
    std::string text;
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lengths;
    std::unordered_map<std::string, int32_t> tid_of;
    uint64_t first_record = 0;
    struct RefIndex {
        std::unordered_map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        std::vector<uint64_t> linear;
    };
    std::vector<RefIndex> index;
    bool has_index = false;
    // CPU seconds per unit of the summariser's last calls on this file (0: none yet): sizes the next call's burst
    mutable std::atomic<double> cpu_s_per_unit{0.0};
};

namespace {

void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t>& bins)
{
    --end;
    bins.clear();
    bins.push_back(0);
    const int shifts[5] = {26, 23, 20, 17, 14};
    const uint32_t offs[5] = {1, 9, 73, 585, 4681};
    for (int l = 0; l < 5; ++l)
        for (int64_t k = offs[l] + (beg >> shifts[l]); k <= (int64_t)offs[l] + (end >> shifts[l]); ++k) bins.push_back((uint32_t)k);
}

inline uint32_t le32(const uint8_t* d) { return (uint32_t)d[0] | (d[1] << 8) | (d[2] << 16) | ((uint32_t)d[3] << 24); }

// The bytes of the next alignment (after its length word): in place inside the inflated block when the
// record does not straddle a block boundary -- no copy, which is what makes walking up to a window cheap
// -- otherwise gathered into `buf`.  nullptr at the end of the data / on a bad length.
const uint8_t* next_record(Bgzf& z, std::vector<uint8_t>& buf, uint32_t& size)
{
    constexpr uint32_t kMaxRecord = 1u << 28;   // no alignment record is a quarter of a gigabyte: a corrupt length
    if (const uint8_t* h = z.contiguous(4)) {
        size = le32(h);
        if (size < 32 || size > kMaxRecord) { z.mark_bad(); return nullptr; }
        if (const uint8_t* d = z.contiguous(4 + (size_t)size)) {
            z.advance(4 + (size_t)size);
            return d + 4;
        }
    }
    uint8_t szb[4];
    if (z.read(szb, 4) != 4) return nullptr;
    size = le32(szb);
    if (size < 32 || size > kMaxRecord) { z.mark_bad(); return nullptr; }
    buf.resize(size);
    if (z.read(buf.data(), size) != size) return nullptr;
    return buf.data();
}

struct RecordLayout { unsigned l_name = 0, n_cigar = 0; size_t tags_off = 0; };

// fixed fields + reference end: all a fetch needs to decide whether the record overlaps its window
bool decode_core(const uint8_t* d, uint32_t size, Record& r, RecordLayout& lay)
{
    r.tid = (int32_t)le32(d);
    r.pos = (int32_t)le32(d + 4);
    lay.l_name = d[8];
    r.mapq = d[9];
    lay.n_cigar = d[12] | (d[13] << 8);
    r.flag = (uint16_t)(d[14] | (d[15] << 8));
    r.l_seq = (int32_t)le32(d + 16);
    r.tlen = (int32_t)le32(d + 28);
    size_t off = 32;
    if (off + lay.l_name + 4ull * lay.n_cigar > size) return false;
    off += lay.l_name;
    r.end = r.pos;
    for (unsigned k = 0; k < lay.n_cigar; ++k) {
        const uint32_t c = le32(d + off + 4 * k);
        if (consumes_ref((int)(c & 0xF))) r.end += (int64_t)(c >> 4);
    }
    off += 4ull * lay.n_cigar;
    off += (size_t)((r.l_seq + 1) / 2 + r.l_seq);
    if (off > size) return false;
    lay.tags_off = off;
    return true;
}

// the variable-length parts a kept record is asked for: query name, CIGAR operations, tag area
void decode_rest(const uint8_t* d, uint32_t size, const RecordLayout& lay, Record& r)
{
    r.name = reinterpret_cast<const char*>(d + 32);
    r.name_len = lay.l_name ? lay.l_name - 1 : 0;

    r.cigar.clear();
    const uint8_t* c0 = d + 32 + lay.l_name;
    for (unsigned k = 0; k < lay.n_cigar; ++k) {
        const uint32_t c = le32(c0 + 4 * k);
        r.cigar.emplace_back((int)(c & 0xF), (int64_t)(c >> 4));
    }
    r.tags = d + lay.tags_off;
    r.tags_len = size - lay.tags_off;
}

bool read_record(Bgzf& z, std::vector<uint8_t>& buf, Record& r)
{
    uint32_t size = 0;
    const uint8_t* d = next_record(z, buf, size);
    RecordLayout lay;
    if (!d || !decode_core(d, size, r, lay)) return false;
    decode_rest(d, size, lay, r);
    return true;
}

// pysam-style fetch: records with pos < end and reference end > beg, in file order; `fn` returns
// false to stop.  Mirrors svtyper_amd/bam.py::AlignmentFile.fetch.
template <typename Fn>
bool fetch(const svt_bam& bam, Bgzf& z, int32_t tid, int64_t beg, int64_t end, std::vector<uint8_t>& buf, Fn&& fn)
{
    if (tid < 0 || tid >= (int32_t)bam.ref_names.size()) return false;
    beg = std::max<int64_t>(beg, 0);
    if (end <= beg) return true;
    const auto& ri = bam.index[tid];
    uint64_t min_off = 0;
    const size_t li = (size_t)(beg >> 14);
    if (!ri.linear.empty()) min_off = li < ri.linear.size() ? ri.linear[li] : ri.linear.back();
    // (scratch of the calling thread, reused from fetch to fetch: two fetches per unit, three allocations each)
    static thread_local std::vector<uint32_t> bins;
    static thread_local std::vector<std::pair<uint64_t, uint64_t>> chunks, merged;
    reg2bins(beg, end, bins);
    chunks.clear();
    merged.clear();
    for (uint32_t b : bins) {
        auto it = ri.bins.find(b);
        if (it == ri.bins.end()) continue;
        for (const auto& c : it->second)
            if (c.second > min_off) chunks.push_back(c);
    }
    if (chunks.empty()) return true;
    std::sort(chunks.begin(), chunks.end());
    merged.push_back(chunks[0]);
    for (size_t i = 1; i < chunks.size(); ++i) {
        if (chunks[i].first <= merged.back().second) merged.back().second = std::max(merged.back().second, chunks[i].second);
        else merged.push_back(chunks[i]);
    }
    Record r;
    for (const auto& c : merged) {
        z.seek(c.first);
        while (z.tell() < c.second) {
            uint32_t size = 0;
            const uint8_t* d = next_record(z, buf, size);
            RecordLayout lay;
            if (!d || !decode_core(d, size, r, lay)) break;
            if (r.tid != tid || r.pos >= end) return true;
            int64_t rend = r.end;
            if (lay.n_cigar == 0 || rend <= r.pos) rend = r.pos + 1;
            if (rend > beg) {      // most records walked on the way to the window stop here, undecoded
                decode_rest(d, size, lay, r);
                if (!fn(r)) return true;
            }
        }
    }
    return !z.failed();
}

// SplitRead.is_valid (parsers.py:959-1058 / fragments.py) -> fills `out` when the candidate is valid
// returns 1 valid, 0 invalid, -1 malformed input
// `sa_seen` / `tags_from`: what the caller's search for RG already knows -- an SA value it walked past, else where the tags it
// has not looked at begin (0: the whole tag area)
int split_candidate(const svt_bam& bam, const Record& r, Split& out, const char* sa_seen = nullptr, size_t tags_from = 0)
{
    bool malformed = false;
    const char* sa = sa_seen;
    if (sa_seen) (void)find_z_tag(r, 0, 0, &malformed, tags_from, nullptr, 0, 0, nullptr, /*validate_only=*/true);   // (the full walk looked at every tag)
    else {
        size_t behind_sa = 0;
        sa = find_z_tag(r, 'S', 'A', &malformed, tags_from, &behind_sa);
        if (sa && !malformed) (void)find_z_tag(r, 0, 0, &malformed, behind_sa, nullptr, 0, 0, nullptr, /*validate_only=*/true);   // (the tags behind SA, too)
    }
    if (malformed) return -1;
    if (r.cigar.empty()) return 0;   // a mapped read without a CIGAR cannot be a split candidate (fragments.py: add_read)
    if (!sa) {   // the common read: no SA tag and no clipped end -> not a candidate, nothing to build
        if (!is_clip(r.cigar.front().first) && !is_clip(r.cigar.back().first)) return 0;
    }
    Piece a;
    a.tid = r.tid;
    a.start = r.pos;
    a.end = r.end;
    a.reverse = (r.flag & 0x10) != 0;
    a.mapq = r.mapq;
    a.cigar = &r.cigar;
    a.qp = query_pos_from_cigar(*a.cigar, a.reverse);
    if (!sa) {
        const bool fc = is_clip(r.cigar.front().first), lc = is_clip(r.cigar.back().first);
        if (!(fc || lc)) return 0;
        const int64_t clip_length = std::max(r.cigar.front().second * (fc ? 1 : 0), r.cigar.back().second * (lc ? 1 : 0));
        int64_t q_aln = 0;
        for (const auto& c : r.cigar) if (consumes_query(c.first)) q_aln += c.second;
        if (clip_length > 0 && (r.l_seq - q_aln) <= 50) {
            Piece dummy;
            dummy.tid = -2;
            dummy.start = 1;
            dummy.end = 1;
            dummy.reverse = a.reverse;
            dummy.mapq = 0;
            dummy.cigar = &r.cigar;
            dummy.qp = query_pos_from_cigar(*dummy.cigar, dummy.reverse);
            out.soft = true;
            if (left_clipped(*a.cigar)) { out.left = dummy; out.right = a; }
            else { out.left = a; out.right = dummy; }
            return 1;
        }
        return 0;
    }
    // SA:Z:chrom,pos,strand,CIGAR,mapQ,NM;...   more than one entry -> discarded (:992-993).  Fields are cut in place (no
    // string per field: a read with an SA tag used to cost a dozen allocations here)
    size_t len = std::strlen(sa);
    while (len && sa[len - 1] == ';') --len;
    if (std::memchr(sa, ';', len)) return 0;
    const char* fb[8];
    size_t fl[8];
    size_t n_fld = 0;
    for (size_t p0 = 0;;) {
        const char* c = static_cast<const char*>(std::memchr(sa + p0, ',', len - p0));
        const size_t p1 = c ? (size_t)(c - sa) : len;
        if (n_fld < 8) { fb[n_fld] = sa + p0; fl[n_fld] = p1 - p0; }
        ++n_fld;
        if (!c) break;
        p0 = p1 + 1;
    }
    if (n_fld < 5) return -1;
    auto whole_number = [](const char* b, size_t n, long long& v) {      // strtoll over the whole field, as before
        char buf[32];
        if (n == 0 || n >= sizeof buf) return false;
        std::memcpy(buf, b, n);
        buf[n] = 0;
        char* endp = nullptr;
        v = std::strtoll(buf, &endp, 10);
        return *endp == 0;
    };
    long long mate_pos1 = 0, mate_mapq = 0;
    if (!whole_number(fb[1], fl[1], mate_pos1)) return -1;
    if (!whole_number(fb[4], fl[4], mate_mapq)) return -1;
    Piece b;
    const std::string sa_chrom(fb[0], fl[0]);          // (chromosome names fit the small-string buffer)
    auto it = bam.tid_of.find(sa_chrom);
    b.tid = it == bam.tid_of.end() ? -3 : it->second;
    b.start = mate_pos1 - 1;
    b.reverse = fl[2] == 1 && fb[2][0] == '-';
    static thread_local Cigar sa_cigar;                // the SA entry's operations: scratch of this thread, alive while `out` is read
    sa_cigar.clear();
    if (!parse_cigar_string(fb[3], fl[3], sa_cigar)) return -1;
    b.cigar = &sa_cigar;
    b.mapq = mate_mapq;
    b.end = b.start;
    for (const auto& c : sa_cigar) if (consumes_ref(c.first)) b.end += c.second;
    b.qp = query_pos_from_cigar(sa_cigar, b.reverse);
    const bool same_chrom = r.tid >= 0 && bam.ref_names[r.tid] == sa_chrom;
    out.soft = false;
    if (same_chrom) {
        if (r.pos > b.start) { out.left = b; out.right = a; }
        else { out.left = a; out.right = b; }
    } else if (a.cigar->empty()) {
        return -1;
    } else if (left_clipped(*a.cigar)) {
        out.left = b; out.right = a;
    } else {
        out.left = a; out.right = b;
    }
    const QueryPos &l = out.left.qp, &rq = out.right.qp;
    const int64_t shared = std::max<int64_t>(0, 1 + std::min(l.end, rq.end) - std::max(l.start, rq.start));
    const int64_t non_overlap = std::min(1 + l.end - l.start - shared, 1 + rq.end - rq.start - shared);
    if (non_overlap < 20) return 0;
    if (out.left.tid == out.right.tid && out.left.reverse == out.right.reverse) {
        auto start_diag = [](const Piece& p) { return p.start - (p.reverse ? p.qp.length - p.qp.end : p.qp.start); };
        auto end_diag = [](const Piece& p) { return p.end - (p.reverse ? p.qp.length - p.qp.start : p.qp.end); };
        const int64_t ins = out.left.reverse ? end_diag(out.right) - start_diag(out.left)
                                             : end_diag(out.left) - start_diag(out.right);
        if (std::llabs(ins) < 50) return 0;
        const int64_t desert = rq.start - l.end - 1;
        if (desert > 0 && desert - std::max<int64_t>(0, ins) > 50) return 0;
    }
    return 1;
}

// Maximal gap-free aligned reference intervals of a read (geometry.aligned_intervals), reduced to what a
// summary keeps: all of them when there are at most two, else the two closest to the unit's breakends in
// the order of a stable sort by distance (geometry._read_words).
void aligned_intervals(const Record& r, int64_t near_a, int64_t near_b, std::vector<std::pair<int64_t, int64_t>>& scratch, ReadInfo& out)
{
    scratch.clear();
    int64_t p = r.pos;
    bool open = false;
    for (const auto& c : r.cigar) {
        if (is_aligned(c.first)) {
            if (!open) { scratch.emplace_back(p, p + c.second); open = true; }
            else scratch.back().second = p + c.second;
            p += c.second;
        } else if (c.first == 2 || c.first == 3) {
            open = false;
            p += c.second;
        }
    }
    if (scratch.size() > 2) {
        auto dist = [&](const std::pair<int64_t, int64_t>& iv) {
            auto one = [&](int64_t q) { return (iv.first <= q && q <= iv.second) ? (int64_t)0 : std::min(std::llabs(iv.first - q), std::llabs(iv.second - q)); };
            return std::min(one(near_a), one(near_b));
        };
        std::stable_sort(scratch.begin(), scratch.end(), [&](const auto& x, const auto& y) { return dist(x) < dist(y); });
        scratch.resize(2);
    }
    out.n_iv = (int)scratch.size();
    for (int k = 0; k < out.n_iv; ++k) {
        out.iv_start[k] = scratch[(size_t)k].first;
        out.iv_end[k] = scratch[(size_t)k].second;
    }
}

inline int32_t clip32(int64_t x) { return (int32_t)std::max<int64_t>(INT32_MIN, std::min<int64_t>(INT32_MAX, x)); }

void fill_read(svt_read_summary& d, const ReadInfo& r)
{
    d.tid = r.tid;
    d.start = clip32(r.start);
    d.end = clip32(r.end);
    for (int k = 0; k < r.n_iv; ++k) {
        d.iv_start[k] = clip32(r.iv_start[k]);
        d.iv_end[k] = clip32(r.iv_end[k]);
    }
    d.mapq = (uint8_t)r.mapq;
    d.flags = (uint8_t)(SVT_READ_PRESENT | (r.reverse ? SVT_READ_REVERSE : 0));
}

inline PieceOut piece_out(const Piece& p)
{
    PieceOut o;
    o.tid = p.tid;
    o.start = p.start;
    o.end = p.end;
    o.mapq = p.mapq;
    o.reverse = p.reverse;
    return o;
}

bool fill_piece(svt_piece_summary& d, const PieceOut& p)
{
    if (p.mapq < 0) return false;
    d.tid = p.tid;
    d.start = clip32(p.start);
    d.end = clip32(p.end);
    d.mapq = (uint8_t)std::min<int64_t>(p.mapq, 255);   // an SA-tag MAPQ above 255: prob_mapq is exactly 1.0 from 163 on (packer.py: _mapq)
    d.flags = (uint8_t)(SVT_READ_PRESENT | (p.reverse ? SVT_READ_REVERSE : 0));
    return true;
}

struct UnitOut {                       // per worker, reused for every unit it processes
    std::vector<svt_fragment> frags;
    std::vector<svt_record> recs;      // svt_bam_evidence: the summaries turned into evidence records
    bool skipped = false;
};

// Per-worker scratch of process_unit, reused from unit to unit so that a read costs no allocation: the
// read-fragments of the unit (query name -> Fragment) live in a vector indexed through an open-addressing
// hash table over a name arena, and are emitted in sorted(query_name) order at the end.
struct Workspace {
    std::vector<Fragment> frags;       // [0, n_frags) are live
    size_t n_frags = 0;
    std::vector<char> names;
    std::vector<uint64_t> table;       // (name hash's high half) << 32 | fragment index + 1, 0 = empty; size is a power of two
    std::vector<uint32_t> order;
    std::vector<std::pair<uint64_t, uint32_t>> keys;
    std::vector<uint64_t> packed;
    std::vector<std::pair<int64_t, int64_t>> intervals;
    std::vector<const SplitOut*> seq, clip;
    std::string last_rg;               // most reads of a unit share their read group
    int32_t last_lib = 0;
    bool have_last_rg = false;

    void begin_unit()
    {
        n_frags = 0;
        names.clear();
        if (table.size() < 1024) table.assign(1024, 0ull);
        else std::fill(table.begin(), table.end(), 0ull);
    }
    static uint64_t hash_name(const char* p, size_t n)
    {
        // eight bytes per step: a byte-wise FNV-1a is a serial chain of one multiply per byte of a 20-50 byte name (a tenth of
        // what a kept read costs)
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t w;
            std::memcpy(&w, p + i, 8);
            h = (h ^ w) * 0xFF51AFD7ED558CCDull;
            h ^= h >> 32;
        }
        if (i < n) {
            uint64_t w = 0;
            std::memcpy(&w, p + i, n - i);
            h = (h ^ w) * 0xFF51AFD7ED558CCDull;
            h ^= h >> 32;
        }
        return h;
    }
    const char* name_of(const Fragment& f) const { return names.data() + f.name_off; }
    // the fragment of this query name; created (with `lib`) when the name is new
    Fragment& fragment(const char* name, const uint32_t len, int lib)
    {
        if ((n_frags + 1) * 2 > table.size()) grow();
        const size_t mask = table.size() - 1;
        const uint64_t h = hash_name(name, len), tag = h & 0xffffffff00000000ull;
        // the slot carries the hash's high half: a probe that meets another name's slot moves on without touching that
        // fragment (a big struct) or its name; the second read of a pair pays ONE memcmp
        for (size_t i = h & mask;; i = (i + 1) & mask) {
            const uint64_t e = table[i];
            if (e == 0ull) {
                if (n_frags == frags.size()) frags.emplace_back();
                Fragment& f = frags[n_frags];
                f.reset(lib, (uint32_t)names.size(), len);
                names.insert(names.end(), name, name + len);
                table[i] = tag | (uint64_t)++n_frags;
                return f;
            }
            if ((e & 0xffffffff00000000ull) != tag) continue;
            Fragment& f = frags[(uint32_t)e - 1];
            if (f.name_len == len && std::memcmp(name_of(f), name, len) == 0) return f;
        }
    }
    void grow()
    {
        table.assign(table.size() * 2, 0ull);
        const size_t mask = table.size() - 1;
        for (size_t k = 0; k < n_frags; ++k) {
            const uint64_t h = hash_name(name_of(frags[k]), frags[k].name_len);
            size_t i = h & mask;
            while (table[i]) i = (i + 1) & mask;
            table[i] = (h & 0xffffffff00000000ull) | (uint64_t)(k + 1);
        }
    }
    // live fragments in the order of Python's sorted() over their (ASCII) names.  Query names of one run share a long
    // prefix (instrument : run : flowcell : lane ...), so comparing them byte by byte from the start -- a few hundred
    // times per unit -- reads the same thirty bytes again and again: the common prefix of the unit's names is found once
    // and the sort runs on the eight bytes behind it as one big-endian integer; equal keys fall back to the whole names.
    const std::vector<uint32_t>& sorted_order()
    {
        order.resize(n_frags);
        keys.resize(n_frags);
        size_t lcp = n_frags ? frags[0].name_len : 0;
        for (size_t k = 1; k < n_frags && lcp; ++k) {
            const char *a = name_of(frags[0]), *b = name_of(frags[k]);
            const size_t n = std::min<size_t>(lcp, frags[k].name_len);
            size_t i = 0;
            for (; i + 8 <= n; i += 8) {      // eight bytes at a time: thirty common bytes times a few hundred names per unit
                uint64_t x, y;
                std::memcpy(&x, a + i, 8);
                std::memcpy(&y, b + i, 8);
                if (x != y) { i += (size_t)__builtin_ctzll(x ^ y) >> 3; break; }     // (little-endian: the lowest differing byte)
            }
            while (i < n && a[i] == b[i]) ++i;      // (the tail; at once over when the words differed)
            lcp = i;
        }
        auto key_of = [&](const Fragment& f) {
            uint64_t key = 0;                                    // bytes past the end count as 0: a shorter name sorts first,
            const unsigned char* p = reinterpret_cast<const unsigned char*>(name_of(f)) + lcp;   // as it does for memcmp + length
            const size_t have = f.name_len - lcp;               // (lcp <= every name's length)
            if (have >= 8) {
                std::memcpy(&key, p, 8);
                return __builtin_bswap64(key);
            }
            for (size_t i = 0; i < 8; ++i) key = (key << 8) | (i < have ? p[i] : 0u);
            return key;
        };
        auto by_name = [&](const uint32_t x, const uint32_t y) {
            const Fragment &a = frags[x], &b = frags[y];
            const int c = std::memcmp(name_of(a), name_of(b), std::min(a.name_len, b.name_len));
            return c != 0 ? c < 0 : a.name_len < b.name_len;
        };
        if (n_frags <= 4096) {
            // the usual unit: the key's leading 52 bits and the fragment's index in ONE integer -- a sort of plain 64-bit words, no
            // comparator that looks at the names; runs of equal leading bits (rare: they agree in six and a half bytes behind the
            // common prefix) are put in order by their whole names afterwards
            packed.resize(n_frags);
            for (size_t k = 0; k < n_frags; ++k) packed[k] = (key_of(frags[k]) & ~uint64_t(0xfff)) | (uint64_t)k;
            std::sort(packed.begin(), packed.end());
            for (size_t k = 0; k < n_frags; ++k) order[k] = (uint32_t)(packed[k] & 0xfffu);
            for (size_t k = 0; k < n_frags;) {
                size_t e = k + 1;
                while (e < n_frags && (packed[e] >> 12) == (packed[k] >> 12)) ++e;
                if (e - k > 1) std::sort(order.begin() + (ptrdiff_t)k, order.begin() + (ptrdiff_t)e, by_name);
                k = e;
            }
            return order;
        }
        for (size_t k = 0; k < n_frags; ++k) keys[k] = std::make_pair(key_of(frags[k]), (uint32_t)k);
        std::sort(keys.begin(), keys.end(), [&](const std::pair<uint64_t, uint32_t>& x, const std::pair<uint64_t, uint32_t>& y) {
            if (x.first != y.first) return x.first < y.first;
            return by_name(x.second, y.second);
        });
        for (size_t k = 0; k < n_frags; ++k) order[k] = keys[k].second;
        return order;
    }
};

struct UnitSpan {                      // where a finished unit's summaries (or evidence records) wait for the gather
    const void* data = nullptr;
    uint64_t count = 0;
    bool skipped = false;
};

// Large host buffers of the summariser (the workers' arenas, the flat summary array): anonymous mappings advised for
// transparent huge pages -- the summaries are written once and read once, so what they cost is page faults and, when they
// go, the unmapping: on the 2 x EPYC 9575F box 17 ms to unmap the arenas of 2.1 M summaries and as much again for the
// flat array, a third of the call.  Mappings therefore go back to a process-wide pool (at most SVT_READER_POOL_MB, default
// 1024, of idle memory; 0 = unmap at once) and the next call starts on pages that are already there.
class BufferPool {
public:
    static BufferPool& get()
    {
        static BufferPool* pool = new BufferPool();      // (never destroyed: buffers may be returned during process exit)
        return *pool;
    }
    // a mapping of at least `bytes` (its real size goes to *cap), nullptr when the system has none
    void* acquire(size_t bytes, size_t* cap)
    {
        const size_t want = (std::max<size_t>(bytes, 1) + kGrain - 1) / kGrain * kGrain;
        {
            std::lock_guard<std::mutex> g(lock_);
            size_t best = idle_.size();
            for (size_t i = 0; i < idle_.size(); ++i)    // smallest idle mapping that fits and is not more than twice too big
                if (idle_[i].second >= want && idle_[i].second <= 2 * want && (best == idle_.size() || idle_[i].second < idle_[best].second)) best = i;
            if (best != idle_.size()) {
                void* p = idle_[best].first;
                *cap = idle_[best].second;
                idle_bytes_ -= *cap;
                idle_.erase(idle_.begin() + (long)best);
                return p;
            }
        }
        void* p = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return nullptr;
        madvise(p, want, MADV_HUGEPAGE);
        *cap = want;
        return p;
    }
    void release(void* p, size_t cap)
    {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(lock_);
            if (idle_bytes_ + cap <= limit_) {
                idle_.emplace_back(p, cap);
                idle_bytes_ += cap;
                return;
            }
        }
        munmap(p, cap);
    }
    // the flat array handed to the caller: its size is remembered here so that svt_summaries_free needs only the pointer
    void* acquire_tracked(size_t bytes)
    {
        size_t cap = 0;
        void* p = acquire(bytes, &cap);
        if (p) {
            std::lock_guard<std::mutex> g(lock_);
            lent_[p] = cap;
        }
        return p;
    }
    // unmap every idle mapping (svt_trim): a long-lived embedding process gives the pool's memory back
    void trim()
    {
        std::vector<std::pair<void*, size_t>> idle;
        {
            std::lock_guard<std::mutex> g(lock_);
            idle.swap(idle_);
            idle_bytes_ = 0;
        }
        for (const auto& m : idle) munmap(m.first, m.second);
    }
    bool release_tracked(void* p)
    {
        size_t cap = 0;
        {
            std::lock_guard<std::mutex> g(lock_);
            auto it = lent_.find(p);
            if (it == lent_.end()) return false;
            cap = it->second;
            lent_.erase(it);
        }
        release(p, cap);
        return true;
    }

private:
    BufferPool()
    {
        if (const char* e = std::getenv("SVT_READER_POOL_MB")) limit_ = (size_t)std::max(0ll, std::atoll(e)) << 20;
    }
    static constexpr size_t kGrain = 2u << 20;           // one huge page
    std::mutex lock_;
    std::vector<std::pair<void*, size_t>> idle_;
    std::unordered_map<void*, size_t> lent_;
    size_t idle_bytes_ = 0, limit_ = (size_t)1024 << 20;
};

// append-only store of one worker: units are copied in whole, never split across chunks
class SummaryArena {
public:
    SummaryArena() = default;
    SummaryArena(const SummaryArena&) = delete;
    SummaryArena& operator=(const SummaryArena&) = delete;
    ~SummaryArena() { for (auto& c : chunks_) BufferPool::get().release(c.first, c.second); }
    const void* append(const void* data, size_t bytes)
    {
        if (bytes == 0) return nullptr;
        if (used_ + bytes > cap_) {
            size_t size = 0;     // 2, 4, 8, 16, 16 ... MiB: forty-seven workers of a small call do not map (and return) 16 MiB each
            void* p = BufferPool::get().acquire(std::max(bytes, std::min(kChunkBytes, (size_t)(2u << 20) << std::min<size_t>(chunks_.size(), 3))), &size);
            if (!p) return nullptr;
            chunks_.emplace_back(p, size);
            cap_ = size;
            used_ = 0;
        }
        uint8_t* dst = static_cast<uint8_t*>(chunks_.back().first) + used_;
        std::memcpy(dst, data, bytes);
        used_ += bytes;
        return dst;
    }

private:
    static constexpr size_t kChunkBytes = 16u << 20;
    std::vector<std::pair<void*, size_t>> chunks_;
    size_t cap_ = 0, used_ = 0;
};

inline double thread_cpu_seconds()
{
    timespec ts;
    return clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0 ? (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec : 0.0;
}

// one unit: gather reads of both windows, assemble fragments, emit summaries
// `emit(fragment)`: what becomes of a finished summary -- kept as it is (svt_bam_summarise) or turned into its 16-byte evidence
// record on the spot (svt_bam_evidence: no array of 128-byte summaries in between); returns false with `err` set to stop
template <typename Emit>
int process_unit(const svt_bam& bam, Bgzf& z, std::vector<uint8_t>& buf, const svt_summarise_args& A,
                 const std::unordered_map<std::string, int32_t>& rg_lib, uint64_t u, Workspace& ws, UnitOut& out,
                 std::string& err, Emit&& emit)
{
    out.frags.clear();
    out.recs.clear();
    out.skipped = false;
    const svt_fetch_unit& w = A.windows[u];
    const int32_t tids[2] = {w.tid_a, w.tid_b};
    const int64_t los[2] = {w.lo_a, w.lo_b}, his[2] = {w.hi_a, w.hi_b};
    const int64_t near_a = A.breakpoints[u].pos_a, near_b = A.breakpoints[u].pos_b;
    ws.begin_unit();
    int rc = SVT_OK;

    // count_mode 1 (singlesample.py:158-185): a unit is skipped when bam.count() of either window exceeds
    // max_reads.  count() looks at the same records the gather pass walks, so the two are one pass here: the
    // reads of a window are counted (pysam's filter: not unmapped / secondary / QC-fail / duplicate) while
    // they are gathered, and the unit is dropped when a window turns out to be over the limit.
    const bool count_windows = A.count_mode == 1 && A.max_reads >= 0;
    for (int s = 0; s < 2 && !out.skipped; ++s) {
        int64_t i = -1, n_counted = 0;
        const bool ok = fetch(bam, z, tids[s], los[s], his[s], buf, [&](const Record& r) {
            ++i;                                                        // enumerate() index of classic.py:79
            if (count_windows && !(r.flag & (0x4 | 0x100 | 0x200 | 0x400)) && ++n_counted > A.max_reads) {
                out.skipped = true;
                return false;
            }
            if (r.flag & (0x4 | 0x400)) return true;                   // unmapped / duplicate
            bool malformed = false;
            size_t behind_rg = 0;
            const char* sa_seen = nullptr;
            const char* rg = find_z_tag(r, 'R', 'G', &malformed, 0, &behind_rg, 'S', 'A', &sa_seen);

            if (malformed || !rg) { err = "read without a usable RG tag: " + r.name_str(); rc = SVT_ERR_INVALID; return false; }
            if (!ws.have_last_rg || ws.last_rg != rg) {
                auto it = rg_lib.find(rg);
                if (it == rg_lib.end()) { err = std::string("read group not in the library table: ") + rg; rc = SVT_ERR_INVALID; return false; }
                ws.last_rg = rg;
                ws.last_lib = it->second;
                ws.have_last_rg = true;
            }
            if (ws.last_lib < 0) return true;                           // library below the prevalence cut
            if (A.count_mode == 0 && A.max_reads >= 0 && i > A.max_reads) { out.skipped = true; return false; }
            Fragment& f = ws.fragment(r.name, r.name_len, ws.last_lib);             // SamFragment(read, lib) when new
            if (std::find(f.seen.begin(), f.seen.end(), r.flag) != f.seen.end()) return true;   // same (name, flag) again
            f.seen.push_back(r.flag);
            if (r.flag & (0x100 | 0x800)) return true;                  // secondary / supplementary
            ReadInfo ri;
            ri.tid = r.tid;
            ri.start = r.pos;
            ri.end = r.end;
            ri.reverse = (r.flag & 0x10) != 0;
            ri.mapq = r.mapq;
            aligned_intervals(r, near_a, near_b, ws.intervals, ri);
            f.primaries.push_back(ri);
            f.num_primary += 1;
            Split sp;
            const int v = split_candidate(bam, r, sp, sa_seen, behind_rg);
            if (v < 0) { err = "malformed SA tag / CIGAR at read " + r.name_str(); rc = SVT_ERR_INVALID; return false; }
            if (v > 0) f.splits.push_back(SplitOut{sp.soft, piece_out(sp.left), piece_out(sp.right)});
            return true;
        });
        if (rc != SVT_OK) return rc;
        if (!ok) { err = "BAM read error"; return SVT_ERR_INVALID; }
    }
    if (out.skipped) { out.frags.clear(); out.recs.clear(); return SVT_OK; }

    for (const uint32_t fi : ws.sorted_order()) {
        const Fragment& f = ws.frags[fi];
        ws.seq.clear();
        ws.clip.clear();
        for (const SplitOut& sp : f.splits) (sp.soft ? ws.clip : ws.seq).push_back(&sp);
        const size_t n_rec = std::max<size_t>({(size_t)1, (f.primaries.size() + 1) / 2, ws.seq.size(), ws.clip.size()});
        for (size_t k = 0; k < n_rec; ++k) {
            svt_fragment fr;
            std::memset(&fr, 0, sizeof fr);
            fr.read[0].tid = fr.read[1].tid = -1;
            for (int j = 0; j < 2; ++j)
                if (2 * k + j < f.primaries.size()) fill_read(fr.read[j], f.primaries[2 * k + j]);
            fr.read[0].reserved = (uint16_t)f.lib;
            fr.read[1].reserved = (uint16_t)(((k == 0 && f.num_primary == 2) ? SVT_FRAG_PAIR : 0) | (k > 0 ? SVT_FRAG_CONTINUATION : 0));
            bool ok = true;
            if (k < ws.seq.size()) ok = fill_piece(fr.seq[0], ws.seq[k]->left) && fill_piece(fr.seq[1], ws.seq[k]->right);
            if (ok && k < ws.clip.size()) ok = fill_piece(fr.clip[0], ws.clip[k]->left) && fill_piece(fr.clip[1], ws.clip[k]->right);
            if (!ok) {
                err = "MAPQ outside 0..255 in an SA tag of fragment " + std::string(ws.name_of(f), f.name_len);
                return SVT_ERR_INVALID;
            }
            if (!emit(fr)) return SVT_ERR_INVALID;
        }
    }
    return SVT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
// svt_trim()'s share of this file: the pooled huge-page buffers of the gather (up to SVT_READER_POOL_MB, 1 GiB by default)
extern "C" void svt_reads_trim() { BufferPool::get().trim(); }      // (internal: not in include/svtyper_reads.h)

extern "C" {

static int svt_bam_open_impl(const char* path, svt_bam** out)
{
    if (!path || !out) return fail(SVT_ERR_INVALID, "null argument");
    *out = nullptr;
    std::unique_ptr<svt_bam> b(new svt_bam());
    b->path = path;
    if (!b->file.open(b->path)) return fail(SVT_ERR_INVALID, std::string("cannot open ") + path);
    Bgzf z(b->file);
    if (!z.ok()) return fail(SVT_ERR_NOMEM, "cannot set up the inflate state");
    uint8_t magic[4];
    z.seek(0);
    auto rd32 = [&](int32_t& v) {
        uint8_t t[4];
        if (z.read(t, 4) != 4) return false;
        v = (int32_t)((uint32_t)t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24));
        return true;
    };
    int32_t l_text = 0, n_ref = 0;
    if (z.read(magic, 4) != 4 || std::memcmp(magic, "BAM\1", 4) != 0 || !rd32(l_text) || l_text < 0)
        return fail(SVT_ERR_INVALID, std::string(path) + " is not a BAM file");
    b->text.resize((size_t)l_text);
    if (l_text && z.read(&b->text[0], (size_t)l_text) != (size_t)l_text) return fail(SVT_ERR_INVALID, "truncated BAM header");
    b->text = b->text.c_str();   // cut at the first NUL
    if (!rd32(n_ref) || n_ref < 0) return fail(SVT_ERR_INVALID, "truncated BAM header");
    for (int32_t i = 0; i < n_ref; ++i) {
        int32_t l_name = 0, l_ref = 0;
        if (!rd32(l_name) || l_name <= 0) return fail(SVT_ERR_INVALID, "truncated BAM header");
        std::string name((size_t)l_name, '\0');
        if (z.read(&name[0], (size_t)l_name) != (size_t)l_name || !rd32(l_ref)) return fail(SVT_ERR_INVALID, "truncated BAM header");
        name.resize((size_t)l_name - 1);
        b->tid_of[name] = i;
        b->ref_names.push_back(name);
        b->ref_lengths.push_back(l_ref);
    }
    b->first_record = z.tell();
    // index: <path>.bai, else the .bai next to the file
    std::string cand[2] = {b->path + ".bai", b->path};
    const size_t dot = cand[1].rfind('.');
    if (dot != std::string::npos) cand[1] = cand[1].substr(0, dot) + ".bai";
    for (const std::string& p : cand) {
        FILE* f = std::fopen(p.c_str(), "rb");
        if (!f) continue;
        std::vector<uint8_t> data;
        uint8_t tmp[65536];
        size_t n;
        while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) data.insert(data.end(), tmp, tmp + n);
        std::fclose(f);
        if (data.size() < 8 || std::memcmp(data.data(), "BAI\1", 4) != 0) return fail(SVT_ERR_INVALID, p + " is not a BAI index");
        size_t off = 4;
        auto u32 = [&](size_t o) { return (uint32_t)data[o] | (data[o + 1] << 8) | (data[o + 2] << 16) | ((uint32_t)data[o + 3] << 24); };
        auto u64 = [&](size_t o) { return (uint64_t)u32(o) | ((uint64_t)u32(o + 4) << 32); };
        const uint32_t nr = u32(off);
        off += 4;
        b->index.resize(nr);
        for (uint32_t r = 0; r < nr; ++r) {
            if (off + 4 > data.size()) return fail(SVT_ERR_INVALID, "truncated BAI");
            const uint32_t n_bin = u32(off);
            off += 4;
            for (uint32_t k = 0; k < n_bin; ++k) {
                if (off + 8 > data.size()) return fail(SVT_ERR_INVALID, "truncated BAI");
                const uint32_t bin = u32(off), n_chunk = u32(off + 4);
                off += 8;
                if (off + 16ull * n_chunk > data.size()) return fail(SVT_ERR_INVALID, "truncated BAI");
                if (bin != 37450) {
                    auto& v = b->index[r].bins[bin];
                    for (uint32_t c = 0; c < n_chunk; ++c) v.emplace_back(u64(off + 16 * c), u64(off + 16 * c + 8));
                }
                off += 16ull * n_chunk;
            }
            if (off + 4 > data.size()) return fail(SVT_ERR_INVALID, "truncated BAI");
            const uint32_t n_intv = u32(off);
            off += 4;
            if (off + 8ull * n_intv > data.size()) return fail(SVT_ERR_INVALID, "truncated BAI");
            for (uint32_t k = 0; k < n_intv; ++k) b->index[r].linear.push_back(u64(off + 8 * k));
            off += 8ull * n_intv;
        }
        b->has_index = true;
        break;
    }
    if (!b->has_index) return fail(SVT_ERR_INVALID, std::string("no .bai index found for ") + path);
    if (b->index.size() < b->ref_names.size()) b->index.resize(b->ref_names.size());
    *out = b.release();
    return SVT_OK;
}

int svt_bam_open(const char* path, svt_bam** out)
{
    return guarded([&] { return svt_bam_open_impl(path, out); });
}

void svt_bam_close(svt_bam* bam) { delete bam; }

int32_t svt_bam_n_references(const svt_bam* bam) { return bam ? (int32_t)bam->ref_names.size() : 0; }

const char* svt_bam_reference_name(const svt_bam* bam, int32_t tid)
{
    return (bam && tid >= 0 && tid < (int32_t)bam->ref_names.size()) ? bam->ref_names[tid].c_str() : nullptr;
}

int64_t svt_bam_reference_length(const svt_bam* bam, int32_t tid)
{
    return (bam && tid >= 0 && tid < (int32_t)bam->ref_lengths.size()) ? bam->ref_lengths[tid] : -1;
}

int32_t svt_bam_tid(const svt_bam* bam, const char* name)
{
    if (!bam || !name) return -1;
    auto it = bam->tid_of.find(name);
    return it == bam->tid_of.end() ? -1 : it->second;
}

const char* svt_bam_header_text(const svt_bam* bam) { return bam ? bam->text.c_str() : nullptr; }

// What the workers' results are gathered into: the three arrays of svt_summaries (elements: 128-byte summaries) or of
// svt_evidence (elements: 16-byte records made from the summaries by svt_geometry_math.h, `geometry` != nullptr).
struct GatherOut {
    uint64_t** offset;
    void** elements;
    uint8_t** skipped;
    size_t element_bytes;
};

static void free_gathered(uint64_t*& offset, void*& elements, uint8_t*& skipped)
{
    std::free(offset);
    if (elements && !BufferPool::get().release_tracked(elements)) std::free(elements);
    std::free(skipped);
    offset = nullptr;
    elements = nullptr;
    skipped = nullptr;
}

static int summarise_units(const svt_bam* bam, const svt_summarise_args* args, const svt_evidence_params* geometry, GatherOut out)
{
    if (!bam || !args) return fail(SVT_ERR_INVALID, "null argument");
    *out.offset = nullptr;
    *out.elements = nullptr;
    *out.skipped = nullptr;
    if (geometry && (geometry->n_libs == 0 || geometry->n_libs > 65536 || !geometry->lib_flank))
        return fail(SVT_ERR_INVALID, "n_libs must be 1..256 with a flank per library");
    const uint64_t n = args->n_units;
    if (n && (!args->windows || !args->breakpoints)) return fail(SVT_ERR_INVALID, "null unit arrays");
    std::unordered_map<std::string, int32_t> rg_lib;
    for (uint32_t i = 0; i < args->n_read_groups; ++i) rg_lib[args->read_groups[i]] = args->read_group_lib[i];

    const bool trace = std::getenv("SVT_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (trace)
            std::fprintf(stderr, "[svt_bam_summarise] %-10s %8.1f ms\n", what,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    std::vector<UnitSpan> outs(n);
    // by default one usable CPU is left to the caller's other thread (the drivers parse the next chunk of the VCF while this
    // runs: pipeline.ChunkPipeline).  A call whose CPU time fits well inside one period of a cgroup quota is a burst
    // (svt_host_cpus.h) and runs on up to 48 physical cores instead: its CPU time is what the last calls on this file
    // measured per unit (+ 30 %), or 350 us per unit when there is none yet -- a window pair at 30x costs 210 us on the
    // 9575F, mostly inflate.  (290 whole-genome-like sites: 97 ms of CPU time, 7.9 -> 2.4 ms; the fixture's 21 100 units:
    // 0.45 s, 31 -> ms -- 16 CPUs for a tenth of a second are the same allowance as 48 for a thirtieth.)
    // (a handle that has not measured anything yet -- every run of a driver opens its own -- goes by what the last call on
    // ANY file of this process measured)
    double known = bam->cpu_s_per_unit.load(std::memory_order_relaxed);
    if (!(known > 0.0)) known = g_cpu_s_per_unit.load(std::memory_order_relaxed);
    const double est_cpu_s = (double)n * (known > 0.0 ? 1.3 * known : 350e-6);
    unsigned nt = args->n_threads > 0 ? (unsigned)args->n_threads : std::max(1u, svt::burst_threads(est_cpu_s, 48u) - 1u);
    nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nt ? nt : 1, n ? n : 1));
    // Consecutive units stay on one worker: neighbouring sites share BGZF blocks, and the worker's own slots serve them
    // without a lock (what a worker re-reads at the start of a run comes from SharedBlocks).  A grab is a long run while
    // there is plenty left and shrinks towards the end, where balance matters: half of an even share of what remains,
    // between 4 (fewer when the call is too small to feed every worker that way) and 64 units (guided self-scheduling).
    const uint64_t min_grab = std::max<uint64_t>(1, std::min<uint64_t>(4, n / (4ull * nt)));
    std::atomic<uint64_t> next(0);
    auto claim = [&](uint64_t& lo, uint64_t& hi) {
        uint64_t at = next.load(std::memory_order_relaxed);
        for (;;) {
            if (at >= n) return false;
            const uint64_t take = std::min<uint64_t>(n - at, std::max<uint64_t>(min_grab, std::min<uint64_t>(64, (n - at) / (2ull * nt))));
            if (next.compare_exchange_weak(at, at + take, std::memory_order_relaxed)) {
                lo = at;
                hi = at + take;
                return true;
            }
        }
    };
    std::atomic<int> first_rc(SVT_OK);
    std::mutex err_lock;
    std::string first_err;
    std::vector<std::unique_ptr<SummaryArena>> arenas(nt);
    const std::unique_ptr<SharedBlocks> shared_blocks(new SharedBlocks());
    struct WorkerStat { double start_s = 0, busy_s = 0, cpu_s = 0, inflate_s = 0; uint64_t units = 0, grabs = 0, inflated = 0, shared = 0, ahead = 0; };
    std::vector<WorkerStat> stats(nt);
    auto worker = [&](unsigned t) {
        const auto w_begin = std::chrono::steady_clock::now();
        stats[t].start_s = std::chrono::duration<double>(w_begin - t_begin).count();
        Bgzf z(bam->file, shared_blocks.get());
        struct Report {
            WorkerStat& st; Bgzf& z; std::chrono::steady_clock::time_point t0; double cpu0;
            ~Report() { st.cpu_s = thread_cpu_seconds() - cpu0; st.busy_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); st.inflate_s = z.inflate_s; st.inflated = z.n_inflated; st.shared = z.n_shared_hits; st.ahead = z.n_ahead; }
        } report{stats[t], z, w_begin, thread_cpu_seconds()};
        std::vector<uint8_t> buf;
        UnitOut unit;
        Workspace ws;
        arenas[t].reset(new SummaryArena());
        if (!z.ok()) {
            std::lock_guard<std::mutex> g(err_lock);
            if (first_rc.exchange(SVT_ERR_NOMEM) == SVT_OK) first_err = "cannot set up the inflate state";
            return;
        }
        for (;;) {   // consecutive units stay on one thread: neighbouring sites share BGZF blocks (and its cache)
            uint64_t u0, u1;
            if (!claim(u0, u1)) return;
            stats[t].grabs += 1;
            stats[t].units += u1 - u0;
            for (uint64_t u = u0; u < u1; ++u) {
                if (first_rc.load(std::memory_order_relaxed) != SVT_OK) return;
                std::string err;
                int rc;
                if (geometry) {      // the predicates of the device stage, here: 16 bytes per fragment leave the reader
                    const svt_breakpoint& bp = args->breakpoints[u];
                    if (bp.svtype > SVT_SVTYPE_BND) { rc = SVT_ERR_INVALID; err = "bad svtype"; }
                    else rc = process_unit(*bam, z, buf, *args, rg_lib, u, ws, unit, err, [&](const svt_fragment& f) {
                        const uint32_t lib = f.read[0].reserved;
                        if (lib >= geometry->n_libs) { err = "library index of a fragment outside the library table"; return false; }
                        const svt::Record4 r = svt::geometry_record(svt::read_of(f.read[0]), svt::read_of(f.read[1]), svt::piece_of(f.seq[0]),
                                                                    svt::piece_of(f.seq[1]), svt::piece_of(f.clip[0]), svt::piece_of(f.clip[1]), bp,
                                                                    geometry->lib_flank[lib], geometry->min_aligned, geometry->split_slop);
                        static_assert(sizeof(svt_record) == sizeof r, "svt_record is four words");
                        unit.recs.emplace_back();
                        std::memcpy(&unit.recs.back(), &r, sizeof r);
                        return true;
                    });
                } else {
                    rc = process_unit(*bam, z, buf, *args, rg_lib, u, ws, unit, err, [&](const svt_fragment& f) { unit.frags.push_back(f); return true; });
                }
                if (rc == SVT_OK) {
                    outs[u].count = geometry ? unit.recs.size() : unit.frags.size();
                    outs[u].skipped = unit.skipped;
                    outs[u].data = geometry ? arenas[t]->append(unit.recs.data(), unit.recs.size() * sizeof(svt_record))
                                            : arenas[t]->append(unit.frags.data(), unit.frags.size() * sizeof(svt_fragment));
                    if (outs[u].count && !outs[u].data) { rc = SVT_ERR_NOMEM; err = "out of host memory"; }
                }
                if (rc != SVT_OK) {
                    std::lock_guard<std::mutex> g(err_lock);
                    if (first_rc.exchange(rc) == SVT_OK) first_err = err;
                    return;
                }
            }
        }
    };
    run_threads(nt, worker);
    if (first_rc.load() != SVT_OK) return fail(first_rc.load(), first_err);
    lap("units");
    {   // what a unit of this file costs: half the last call, half the calls before it
        double cpu = 0.0;
        for (const auto& w : stats) cpu += w.cpu_s;
        if (n && cpu > 0.0) {
            const double now = cpu / (double)n, before = bam->cpu_s_per_unit.load(std::memory_order_relaxed);
            bam->cpu_s_per_unit.store(before > 0.0 ? 0.5 * (before + now) : now, std::memory_order_relaxed);
            g_cpu_s_per_unit.store(now, std::memory_order_relaxed);
        }
        svt::note_cpu_s(cpu);
    }
    if (trace) {
        WorkerStat sum, longest;
        double first_start = 1e9, last_start = 0, first_end = 1e9, last_end = 0;
        for (const auto& w : stats) {
            first_start = std::min(first_start, w.start_s); last_start = std::max(last_start, w.start_s);
            first_end = std::min(first_end, w.start_s + w.busy_s); last_end = std::max(last_end, w.start_s + w.busy_s);
            sum.busy_s += w.busy_s; sum.cpu_s += w.cpu_s; sum.inflate_s += w.inflate_s; sum.inflated += w.inflated; sum.shared += w.shared; sum.grabs += w.grabs; sum.ahead += w.ahead;
            if (w.busy_s > longest.busy_s) longest = w;
        }
        std::fprintf(stderr, "[svt_bam_summarise] workers started %.2f .. %.2f ms, finished %.2f .. %.2f ms\n", first_start * 1e3, last_start * 1e3, first_end * 1e3, last_end * 1e3);
        std::fprintf(stderr, "[svt_bam_summarise] %u workers: CPU %.1f ms, busy %.1f ms in all (longest %.1f ms: %llu units in %llu grabs, %.1f ms inflating), %llu grabs, "
                             "%llu blocks inflated in %.1f ms (%llu of them ahead for others), %llu taken from other workers\n", nt, sum.cpu_s * 1e3, sum.busy_s * 1e3, longest.busy_s * 1e3,
                     (unsigned long long)longest.units, (unsigned long long)longest.grabs, longest.inflate_s * 1e3, (unsigned long long)sum.grabs,
                     (unsigned long long)sum.inflated, sum.inflate_s * 1e3, (unsigned long long)sum.ahead, (unsigned long long)sum.shared);
    }

    uint64_t total = 0;
    for (const auto& o : outs) total += o.count;
    uint64_t* offsets = static_cast<uint64_t*>(std::malloc((n + 1) * sizeof(uint64_t)));
    void* elements = nullptr;
    {   // from the pool of huge-page mappings when it is large (1.4 GB for 10 M summaries), malloc otherwise
        const size_t bytes = std::max<uint64_t>(total, 1) * out.element_bytes;
        elements = bytes >= (4u << 20) ? BufferPool::get().acquire_tracked(bytes) : std::malloc(bytes);
    }
    uint8_t* skipped = static_cast<uint8_t*>(std::malloc(std::max<uint64_t>(n, 1)));
    if (!offsets || !elements || !skipped) {
        free_gathered(offsets, elements, skipped);
        return fail(SVT_ERR_NOMEM, "out of host memory");
    }
    uint64_t off = 0;
    for (uint64_t u = 0; u < n; ++u) {
        offsets[u] = off;
        off += outs[u].count;
        skipped[u] = outs[u].skipped ? 1 : 0;
    }
    offsets[n] = off;
    {   // gather the per-unit vectors into the flat array on the same threads
        std::atomic<uint64_t> nextu(0);
        auto copier = [&]() {
            for (;;) {
                const uint64_t u0 = nextu.fetch_add(256);
                if (u0 >= n) return;
                for (uint64_t u = u0; u < std::min(n, u0 + 256); ++u)
                    if (outs[u].count)
                        std::memcpy(static_cast<uint8_t*>(elements) + offsets[u] * out.element_bytes, outs[u].data, outs[u].count * out.element_bytes);
            }
        };
        run_threads(std::min(nt, 32u), [&](unsigned) { copier(); });
    }
    *out.offset = offsets;
    *out.elements = elements;
    *out.skipped = skipped;
    lap("gather");
    arenas.clear();
    lap("release");
    return SVT_OK;
}

int svt_bam_summarise(const svt_bam* bam, const svt_summarise_args* args, svt_summaries* out)
{
    return guarded([&] {
        if (!out) return fail(SVT_ERR_INVALID, "null argument");
        void* elements = nullptr;
        const int rc = summarise_units(bam, args, nullptr, GatherOut{&out->frag_offset, &elements, &out->skipped, sizeof(svt_fragment)});
        out->fragments = static_cast<svt_fragment*>(elements);
        return rc;
    });
}

void svt_summaries_free(svt_summaries* s)
{
    if (!s) return;
    void* elements = s->fragments;
    free_gathered(s->frag_offset, elements, s->skipped);
    s->fragments = nullptr;
}

int svt_bam_evidence(const svt_bam* bam, const svt_summarise_args* args, const svt_evidence_params* geometry, svt_evidence* out)
{
    return guarded([&] {
        if (!out || !geometry) return fail(SVT_ERR_INVALID, "null argument");
        void* elements = nullptr;
        const int rc = summarise_units(bam, args, geometry, GatherOut{&out->rec_offset, &elements, &out->skipped, sizeof(svt_record)});
        out->records = static_cast<svt_record*>(elements);
        return rc;
    });
}

void svt_evidence_free(svt_evidence* e)
{
    if (!e) return;
    void* elements = e->records;
    free_gathered(e->rec_offset, elements, e->skipped);
    e->records = nullptr;
}

static int svt_bam_scan_library_impl(const svt_bam* bam, uint32_t n_read_groups, const char* const* read_groups, int64_t num_samp,
                         svt_library_scan* out)
{
    if (!bam || !out || (n_read_groups && !read_groups)) return fail(SVT_ERR_INVALID, "null argument");
    *out = svt_library_scan{};
    std::set<std::string> rgset;
    for (uint32_t i = 0; i < n_read_groups; ++i) rgset.insert(read_groups[i]);
    Bgzf z(bam->file);
    if (!z.ok()) return fail(SVT_ERR_NOMEM, "cannot set up the inflate state");
    std::vector<uint8_t> buf;
    Record r;
    // 1 in the set, 0 not in the set, -1 no usable RG tag (an error where the reference calls get_tag)
    auto in_library = [&](const Record& rec) -> int {
        bool malformed = false;
        const char* rg = find_z_tag(rec, 'R', 'G', &malformed);
        if (malformed || !rg) return -1;
        return rgset.count(rg) ? 1 : 0;
    };
    auto no_rg = [&](const Record& rec) { return fail(SVT_ERR_INVALID, "read without a usable RG tag: " + rec.name_str()); };
    auto query_length = [](const Record& rec) {
        int64_t n = 0;
        for (const auto& c : rec.cigar) if (c.first == 0 || c.first == 1 || c.first == 4 || c.first == 7 || c.first == 8) n += c.second;
        return n;
    };

    // calc_read_length (parsers.py:516-528)
    z.seek(bam->first_record);
    // (an indexed file is walked reference by reference, pysam's IteratorRowAllRefs: the unplaced unmapped reads a
    //  coordinate-sorted BAM ends with -- reference id -1 -- are never seen by the reference)
    for (int64_t seen = 0; read_record(z, buf, r) && r.tid >= 0;) {
        const int in = in_library(r);
        if (in < 0) return no_rg(r);
        if (!in) continue;
        out->read_length = std::max(out->read_length, query_length(r));
        if (seen == 10000) break;
        ++seen;
    }
    // calc_insert_hist (parsers.py:534-576)
    // keys in order of first occurrence, like the reference's Counter: its mean / sd are sums in that order
    std::vector<int64_t> hist_keys;
    std::vector<uint64_t> hist_counts;
    std::unordered_map<int64_t, size_t> hist_slot;
    z.seek(bam->first_record);
    for (int64_t n = 0; read_record(z, buf, r) && r.tid >= 0;) {
        if ((r.flag & 0x10) || !(r.flag & 0x20) || (r.flag & (0x4 | 0x8)) || (r.flag & (0x100 | 0x800))) continue;
        if (r.tlen <= 0) continue;
        const int in = in_library(r);
        if (in < 0) return no_rg(r);
        if (!in) continue;
        auto slot = hist_slot.find(r.tlen);
        if (slot == hist_slot.end()) {
            hist_slot.emplace(r.tlen, hist_keys.size());
            hist_keys.push_back(r.tlen);
            hist_counts.push_back(1);
        } else {
            ++hist_counts[slot->second];
        }
        if (++n == num_samp) break;    // parsers.py:571-573: tested after the increment, so -n 0 scans the whole file
    }
    // calc_lib_prevalence (parsers.py:501-513)
    z.seek(bam->first_record);
    while (out->total != 100000 && read_record(z, buf, r) && r.tid >= 0) {
        const int in = in_library(r);
        if (in < 0) return no_rg(r);
        out->in_lib += (uint64_t)in;
        ++out->total;
    }
    if (z.failed()) return fail(SVT_ERR_INVALID, "corrupt BGZF block in " + bam->path);
    out->n_hist = hist_keys.size();
    out->hist_keys = static_cast<int64_t*>(std::malloc(std::max<size_t>(hist_keys.size(), 1) * sizeof(int64_t)));
    out->hist_counts = static_cast<uint64_t*>(std::malloc(std::max<size_t>(hist_keys.size(), 1) * sizeof(uint64_t)));
    if (!out->hist_keys || !out->hist_counts) {
        svt_library_scan_free(out);
        return fail(SVT_ERR_NOMEM, "out of host memory");
    }
    if (!hist_keys.empty()) {
        std::memcpy(out->hist_keys, hist_keys.data(), hist_keys.size() * sizeof(int64_t));
        std::memcpy(out->hist_counts, hist_counts.data(), hist_counts.size() * sizeof(uint64_t));
    }
    return SVT_OK;
}

int svt_bam_scan_library(const svt_bam* bam, uint32_t n_read_groups, const char* const* read_groups, int64_t num_samp, svt_library_scan* out)
{
    return guarded([&] { return svt_bam_scan_library_impl(bam, n_read_groups, read_groups, num_samp, out); });
}

void svt_library_scan_free(svt_library_scan* s)
{
    if (!s) return;
    std::free(s->hist_keys);
    std::free(s->hist_counts);
    s->hist_keys = nullptr;
    s->hist_counts = nullptr;
    s->n_hist = 0;
}

}  // extern "C"
