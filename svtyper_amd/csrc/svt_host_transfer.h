// svt_host_transfer.h -- buffer / handle pools and the pinned staging ring (H2D / D2H)
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_HOST_TRANSFER_H
#define SVT_HOST_TRANSFER_H

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "svt_device_types.h"
#include "svt_error.h"
#include "svt_host_cpus.h"

namespace svt {

constexpr int kMaxDevices = 64;   // per-device pools and rings are indexed by the HIP device number

inline unsigned host_threads()
{
    return std::max(1u, std::min(usable_cpus(), 16u));
}

// run fn(i) for i in [0, n) on up to host_threads() threads
template <typename Fn>
inline void parallel_for(uint64_t n, Fn&& fn)
{
    const unsigned nt = (unsigned)std::min<uint64_t>(host_threads(), n);
    if (nt == 0) return;
    run_threads(nt, [&](unsigned t) { for (uint64_t i = t; i < n; i += nt) fn(i); });
}

// Pool of the large resident device buffers (records, offsets, unit headers, result records): a fresh
// hipMalloc is not only slow by itself, the first kernel that touches the new memory also waits ~10-20 ms
// for the driver's asynchronous clear of it (measured: tools/first_pass_probe.py), so a pipeline that
// creates one batch per chunk would pay that on every chunk.  svt_batch_destroy returns the buffers
// here; svt_trim() releases them.
//
// Large buffers (a batch's records) are not one hipMalloc but one virtual range mapped onto physical chunks of 256 MB
// (hipMemCreate / hipMemMap).  How fast the pass streams its records depends on where in HBM they lie relative to the
// result records it writes: with the records in ONE physical allocation the same batch ran at either of two levels 6-8 %
// apart, decided per pair of allocations and constant while they stay put; over records made of separately allocated
// chunks every trial ran at the fast level (profiles/r03_placement_variance.txt, tools/placement_vmm.hip).
struct DevicePool {
    struct Item { void* p; uint64_t cap; int device; bool chunked; };
    static constexpr size_t kMaxItems = 8;
    static constexpr uint64_t kChunkedMin = 512ull << 20, kChunk = 256ull << 20;
    std::mutex lock;
    std::vector<Item> items;
    std::vector<std::pair<void*, uint64_t>> mapped;   // chunked buffers: virtual base, mapped bytes (guarded by `lock`)
    std::atomic<bool> chunked_available{true};                    // false once alloc_chunked has failed (guarded by `lock` where it matters)
    // `bytes` of device memory as one virtual range over 256 MB physical chunks; false: not available, nothing left behind
    bool alloc_chunked(int device, uint64_t bytes, void** out, uint64_t* cap, uint64_t chunk_bytes = 0)
    {
        static const bool enabled = [] { const char* e = std::getenv("SVT_CHUNKED_BUFFERS"); return !(e && std::atoi(e) == 0); }();
        if (!enabled && !chunk_bytes) return false;   // (measurements)
        const uint64_t kChunk = chunk_bytes ? chunk_bytes : DevicePool::kChunk;   // (measurement hooks choose their own chunk size)
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0 || kChunk % gran) {
            (void)hipGetLastError();
            return false;
        }
        const uint64_t total = (bytes + kChunk - 1) / kChunk * kChunk;
        void* va = nullptr;
        if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
        uint64_t done = 0;
        bool ok = true;
        for (; ok && done < total; done += kChunk) {
            hipMemGenericAllocationHandle_t h;
            ok = hipMemCreate(&h, kChunk, &prop, 0) == hipSuccess;
            if (!ok) break;
            ok = hipMemMap(static_cast<char*>(va) + done, kChunk, 0, h, 0) == hipSuccess;
            (void)hipMemRelease(h);        // (the mapping keeps the chunk alive)
            if (!ok) break;
        }
        if (ok) {
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            ok = hipMemSetAccess(va, total, &acc, 1) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            if (done) (void)hipMemUnmap(va, done);
            (void)hipMemAddressFree(va, total);
            return false;
        }
        {
            std::lock_guard<std::mutex> g(lock);
            mapped.emplace_back(va, total);
        }
        *out = va;
        *cap = total;
        return true;
    }
    // hipFree, or unmap + release of a chunked buffer, on the device the buffer lives on (the caller's current device is restored)
    void release(void* p, int device)
    {
        int before = -1;
        if (hipGetDevice(&before) != hipSuccess) before = -1;
        if (before != device) (void)hipSetDevice(device);
        release_here(p);
        if (before >= 0 && before != device) (void)hipSetDevice(before);
    }
    bool is_chunked_locked(const void* p) const
    {
        for (const auto& m : mapped) if (m.first == p) return true;
        return false;
    }
    void release_here(void* p)
    {
        uint64_t total = 0;
        {
            std::lock_guard<std::mutex> g(lock);
            for (size_t i = 0; i < mapped.size(); ++i)
                if (mapped[i].first == p) {
                    total = mapped[i].second;
                    mapped.erase(mapped.begin() + (long)i);
                    break;
                }
        }
        if (total) {
            (void)hipDeviceSynchronize();          // (hipFree waits for the device too)
            (void)hipMemUnmap(p, total);
            (void)hipMemAddressFree(p, total);
        } else {
            (void)hipFree(p);
        }
    }
    // a buffer of at least `bytes` (best fit, at most 2x + 1 MiB oversized), else a new allocation
    // `records`: the buffer will hold a batch's records (only those are built from chunks: a result buffer may be handed to
    // RCCL or another process, which a plain allocation always allows)
    int get(int device, uint64_t bytes, void** out, uint64_t* cap, bool records = false)
    {
        bytes = std::max<uint64_t>(bytes, 256);
        {
            std::lock_guard<std::mutex> g(lock);
            size_t best = items.size();
            // a chunked (virtual-memory) buffer only ever serves a record request of chunked size, a plain allocation
            // everything else: result records may be handed to RCCL or to another process, and a record buffer must keep
            // the placement it was built for
            const bool want_chunked = records && bytes + bytes / 8 >= kChunkedMin;
            for (size_t i = 0; i < items.size(); ++i)
                if (items[i].device == device && items[i].cap >= bytes && items[i].cap <= 2 * bytes + (1u << 20) &&
                    (items[i].chunked == want_chunked || (want_chunked && !chunked_available)) &&
                    (best == items.size() || items[i].cap < items[best].cap))
                    best = i;
            if (best != items.size()) {
                *out = items[best].p;
                *cap = items[best].cap;
                items.erase(items.begin() + (long)best);
                return SVT_OK;
            }
        }
        const uint64_t want = bytes + bytes / 8;   // room for the next, slightly larger batch
        if (records && want >= kChunkedMin) {
            if (alloc_chunked(device, want, out, cap)) return SVT_OK;
            chunked_available = false;   // (no virtual-memory API, or switched off: plain buffers serve record requests too)
        }
        HIP_TRY(hipMalloc(out, want));
        *cap = want;
        return SVT_OK;
    }
    void put(int device, void* p, uint64_t cap)
    {
        if (!p) return;
        Item drop{nullptr, 0, 0, false};
        {
            std::lock_guard<std::mutex> g(lock);
            items.push_back(Item{p, cap, device, is_chunked_locked(p)});
            if (items.size() > kMaxItems) {   // keep the largest ones
                size_t smallest = 0;
                for (size_t i = 1; i < items.size(); ++i)
                    if (items[i].cap < items[smallest].cap) smallest = i;
                drop = items[smallest];
                items.erase(items.begin() + (long)smallest);
            }
        }
        if (drop.p) release(drop.p, drop.device);
    }
    void trim()
    {
        std::vector<Item> all;
        {
            std::lock_guard<std::mutex> g(lock);
            all.swap(items);
        }
        for (const Item& it : all) release(it.p, it.device);
    }
};
inline DevicePool g_pool;

// Streams, events and the small table buffers of a batch, kept between batches: hipStreamCreate / hipStreamDestroy /
// hipEventCreate / hipMalloc / hipFree of a dozen objects per batch cost ~10 ms of a 45 ms one-shot (hipFree and
// hipStreamDestroy synchronise the device), which a chunked driver pays on every chunk.  Everything handed back
// must be idle.  svt_trim() releases the lot.
struct HandlePool {
    std::mutex lock;
    std::vector<hipStream_t> streams[kMaxDevices];
    std::vector<hipEvent_t> timing_events[kMaxDevices], plain_events[kMaxDevices];
    struct Small { void* p; uint64_t cap; };
    std::vector<Small> smalls[kMaxDevices];
    static int dev()
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
        return d;
    }
    int get_stream(hipStream_t* s)
    {
        const int d = dev();
        {
            std::lock_guard<std::mutex> g(lock);
            if (!streams[d].empty()) { *s = streams[d].back(); streams[d].pop_back(); return SVT_OK; }
        }
        HIP_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
        return SVT_OK;
    }
    void put_stream(hipStream_t s)
    {
        if (!s) return;
        std::lock_guard<std::mutex> g(lock);
        streams[dev()].push_back(s);
    }
    int get_event(hipEvent_t* e, bool timing)
    {
        const int d = dev();
        {
            std::lock_guard<std::mutex> g(lock);
            auto& v = timing ? timing_events[d] : plain_events[d];
            if (!v.empty()) { *e = v.back(); v.pop_back(); return SVT_OK; }
        }
        if (timing) HIP_TRY(hipEventCreate(e));
        else HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
        return SVT_OK;
    }
    void put_event(hipEvent_t e, bool timing)
    {
        if (!e) return;
        std::lock_guard<std::mutex> g(lock);
        (timing ? timing_events : plain_events)[dev()].push_back(e);
    }
    // device buffers of at most 1 MiB (look-up tables, descriptors): sizes rounded up to a power of two
    int get_small(uint64_t bytes, void** p)
    {
        uint64_t cap = 256;
        while (cap < bytes) cap <<= 1;
        const int d = dev();
        {
            std::lock_guard<std::mutex> g(lock);
            for (size_t i = 0; i < smalls[d].size(); ++i)
                if (smalls[d][i].cap == cap) {
                    *p = smalls[d][i].p;
                    smalls[d].erase(smalls[d].begin() + (long)i);
                    return SVT_OK;
                }
        }
        HIP_TRY(hipMalloc(p, cap));
        std::lock_guard<std::mutex> g(lock);
        small_caps.push_back(Small{*p, cap});
        return SVT_OK;
    }
    void put_small(void* p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> g(lock);
        for (const Small& s : small_caps)
            if (s.p == p) { smalls[dev()].push_back(s); return; }
        (void)hipFree(p);   // not one of ours
    }
    std::vector<Small> small_caps;   // every small buffer ever handed out (pointer -> capacity)
    void trim()
    {
        std::lock_guard<std::mutex> g(lock);
        for (int d = 0; d < kMaxDevices; ++d) {
            if (streams[d].empty() && timing_events[d].empty() && plain_events[d].empty() && smalls[d].empty()) continue;
            (void)hipSetDevice(d);
            for (hipStream_t s : streams[d]) (void)hipStreamDestroy(s);
            for (hipEvent_t e : timing_events[d]) (void)hipEventDestroy(e);
            for (hipEvent_t e : plain_events[d]) (void)hipEventDestroy(e);
            for (const Small& s : smalls[d]) {
                (void)hipFree(s.p);
                for (size_t i = 0; i < small_caps.size(); ++i)
                    if (small_caps[i].p == s.p) { small_caps.erase(small_caps.begin() + (long)i); break; }
            }
            streams[d].clear(); timing_events[d].clear(); plain_events[d].clear(); smalls[d].clear();
        }
    }
};
inline HandlePool g_handles;

// Pinned staging rings, one per device (allocated on first use): the callers of one device take turns on its
// ring, callers of different devices (svt_genotype_multi: one host thread per GPU) copy concurrently.
struct StagingRing {
    static constexpr uint64_t kPiece = 64ull << 20;
    static constexpr int kSlots = 3;
    void* buf[kSlots] = {nullptr, nullptr, nullptr};
    std::mutex lock;
    int ensure()
    {
        for (int i = 0; i < kSlots; ++i)
            if (!buf[i] && hipHostMalloc(&buf[i], kPiece, hipHostMallocDefault) != hipSuccess)
                return fail(SVT_ERR_HIP, "hipHostMalloc of the pinned staging ring failed");
        return SVT_OK;
    }
};
inline StagingRing g_rings[kMaxDevices];

// the ring of the calling thread's current device (every entry point has called hipSetDevice)
inline StagingRing& current_ring()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return g_rings[dev];
}

// Host -> device copies of pageable memory through the pinned ring: a few host threads fill one piece
// while the previous piece is on the wire (pinned pieces move at ~56 GB/s; a first hipMemcpy of
// pageable memory stages at ~13 GB/s, tools/h2d_probe.hip).  It also keeps the runtime from pinning the
// caller's (or our own std::vector's) pages for a direct DMA: when such pages are unmapped later, the
// driver's MMU notifier evicts the process's GPU queues and the next kernel starts 10-20 ms late
// (tools/first_pass_probe.py).  One Stager holds the ring for its lifetime; copies are asynchronous on
// `stream`, finish() waits for them.
constexpr uint64_t kDirectCopyMax = 256u << 10;   // below this the runtime's own bounce buffer is used anyway

class Stager {
public:
    explicit Stager(hipStream_t stream) : stream_(stream), ring_(current_ring()), guard_(ring_.lock) {}
    Stager(const Stager&) = delete;
    Stager& operator=(const Stager&) = delete;
    ~Stager()
    {
        (void)hipStreamSynchronize(stream_);   // the ring is reusable once the last piece has left
        for (int i = 0; i < StagingRing::kSlots; ++i) g_handles.put_event(done_[i], false);
    }
    int copy(void* dst, const void* src, uint64_t bytes)
    {
        if (bytes == 0) return SVT_OK;
        if (bytes <= kDirectCopyMax) {
            HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream_));
            return SVT_OK;
        }
        SVT_TRY(ring_.ensure());
        for (int i = 0; i < StagingRing::kSlots; ++i)
            if (!done_[i]) SVT_TRY(g_handles.get_event(&done_[i], false));
        const unsigned nt = std::min(host_threads(), 6u);   // 4-8 threads saturate the host copy
        for (uint64_t off = 0; off < bytes; slot_ = (slot_ + 1) % StagingRing::kSlots) {
            const uint64_t len = std::min(StagingRing::kPiece, bytes - off);
            if (hipEventSynchronize(done_[slot_]) != hipSuccess) return fail(SVT_ERR_HIP, "staging event");
            const char* s0 = static_cast<const char*>(src) + off;
            char* p0 = static_cast<char*>(ring_.buf[slot_]);
            if (len < (4u << 20)) std::memcpy(p0, s0, len);
            else {
                const uint64_t part = ((len + nt - 1) / nt + 4095) & ~uint64_t(4095);
                parallel_for(nt, [&](uint64_t t) {
                    const uint64_t lo = t * part, hi = std::min(len, lo + part);
                    if (lo < hi) std::memcpy(p0 + lo, s0 + lo, hi - lo);
                });
            }
            if (hipMemcpyAsync(static_cast<char*>(dst) + off, p0, len, hipMemcpyHostToDevice, stream_) != hipSuccess ||
                hipEventRecord(done_[slot_], stream_) != hipSuccess)
                return fail(SVT_ERR_HIP, "staged hipMemcpyAsync");
            off += len;
        }
        return SVT_OK;
    }
    int finish()
    {
        HIP_TRY(hipStreamSynchronize(stream_));
        return SVT_OK;
    }

private:
    hipStream_t stream_;
    StagingRing& ring_;
    std::lock_guard<std::mutex> guard_;
    hipEvent_t done_[StagingRing::kSlots] = {nullptr, nullptr, nullptr};
    int slot_ = 0;
};

inline int h2d_staged(void* dst, const void* src, uint64_t bytes, hipStream_t stream)
{
    Stager st(stream);
    SVT_TRY(st.copy(dst, src, bytes));
    return st.finish();
}

// Device -> host through the same pinned ring (results: 128 B per unit).
inline int d2h_staged(void* dst, const void* src, uint64_t bytes, hipStream_t stream)
{
    if (bytes <= kDirectCopyMax) {
        if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SVT_OK;
    }
    StagingRing& ring = current_ring();
    std::lock_guard<std::mutex> guard(ring.lock);
    SVT_TRY(ring.ensure());
    const unsigned nt = std::min(host_threads(), 6u);
    // piece k is copied out of its slot while piece k + 1 is on the wire
    uint64_t off = 0, prev_off = 0, prev_len = 0;
    int slot = 0, prev_slot = -1;
    while (off < bytes || prev_slot >= 0) {
        uint64_t len = 0;
        if (off < bytes) {
            len = std::min(StagingRing::kPiece, bytes - off);
            HIP_TRY(hipMemcpyAsync(ring.buf[slot], static_cast<const char*>(src) + off, len, hipMemcpyDeviceToHost, stream));
        }
        if (prev_slot >= 0) {
            const char* p0 = static_cast<const char*>(ring.buf[prev_slot]);
            char* d0 = static_cast<char*>(dst) + prev_off;
            const uint64_t part = ((prev_len + nt - 1) / nt + 4095) & ~uint64_t(4095);
            parallel_for(nt, [&](uint64_t t) {
                const uint64_t lo = t * part, hi = std::min(prev_len, lo + part);
                if (lo < hi) std::memcpy(d0 + lo, p0 + lo, hi - lo);
            });
        }
        HIP_TRY(hipStreamSynchronize(stream));
        prev_slot = len ? slot : -1;
        prev_off = off;
        prev_len = len;
        off += len;
        slot = (slot + 1) % 2;
    }
    return SVT_OK;
}


}  // namespace svt

#endif  // SVT_HOST_TRANSFER_H
