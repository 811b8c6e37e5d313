// svt_host_transfer.h -- device scratch cache and pinned staging ring (H2D / D2H)
// Internal header of libsvtyper_hip.so (single translation unit: svtyper_hip.hip).
#ifndef SVT_HOST_TRANSFER_H
#define SVT_HOST_TRANSFER_H

#include "svt_host_tiling.h"

namespace svt {

// Device scratch for the canonical records of the batch being created: a 1.6 GB hipMalloc costs
// ~100 ms, so the buffer is kept per device between calls (grow-only; svt_trim() releases it).
struct CsrScratchCache {
    static constexpr int kMaxDevices = 64;
    void* ptr[kMaxDevices] = {};
    uint64_t cap[kMaxDevices] = {};
    std::mutex lock;   // held for the whole svt_batch_create of a device-sharing caller
    int acquire(int device, uint64_t bytes, void** out)
    {
        if (device >= kMaxDevices) return fail(SVT_ERR_INVALID, "device index too large for the scratch cache");
        if (cap[device] < bytes) {
            if (ptr[device]) (void)hipFree(ptr[device]);
            ptr[device] = nullptr;
            cap[device] = 0;
            const uint64_t want = bytes + bytes / 8;   // a little slack for the next, slightly larger batch
            HIP_TRY(hipMalloc(&ptr[device], want));
            cap[device] = want;
        }
        *out = ptr[device];
        return SVT_OK;
    }
    void trim()
    {
        std::lock_guard<std::mutex> g(lock);
        for (int d = 0; d < kMaxDevices; ++d)
            if (ptr[d]) {
                (void)hipSetDevice(d);
                (void)hipFree(ptr[d]);
                ptr[d] = nullptr;
                cap[d] = 0;
            }
    }
};
inline CsrScratchCache g_csr_cache;

// Pinned staging ring shared by all batches of the process (allocated on first use, per device
// context of the first caller; pinned host memory is usable from every device).
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
inline StagingRing g_ring;

// Host -> device copy of a large pageable buffer through the pinned ring: a few host threads fill
// one piece while the previous piece is on the wire (a first hipMemcpy of pageable memory stages at
// ~13 GB/s on this platform; pinned pieces move at ~56 GB/s, tools/h2d_probe.hip).
inline int h2d_staged(void* dst, const void* src, uint64_t bytes, hipStream_t stream)
{
    if (bytes < (16ull << 20)) {
        if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        return SVT_OK;
    }
    std::lock_guard<std::mutex> guard(g_ring.lock);
    SVT_TRY(g_ring.ensure());
    hipEvent_t done[StagingRing::kSlots] = {nullptr, nullptr, nullptr};
    int rc = SVT_OK;
    for (int i = 0; i < StagingRing::kSlots && rc == SVT_OK; ++i)
        if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) rc = fail(SVT_ERR_HIP, "hipEventCreate");
    const unsigned nt = std::min(host_threads(), 6u);   // 4-8 threads saturate the host copy
    uint64_t off = 0;
    for (int slot = 0; rc == SVT_OK && off < bytes; slot = (slot + 1) % StagingRing::kSlots) {
        const uint64_t len = std::min(StagingRing::kPiece, bytes - off);
        if (hipEventSynchronize(done[slot]) != hipSuccess) { rc = fail(SVT_ERR_HIP, "staging event"); break; }
        const char* s0 = static_cast<const char*>(src) + off;
        char* p0 = static_cast<char*>(g_ring.buf[slot]);
        const uint64_t part = ((len + nt - 1) / nt + 4095) & ~uint64_t(4095);
        parallel_for(nt, [&](uint64_t t) {
            const uint64_t lo = t * part, hi = std::min(len, lo + part);
            if (lo < hi) std::memcpy(p0 + lo, s0 + lo, hi - lo);
        });
        if (hipMemcpyAsync(static_cast<char*>(dst) + off, p0, len, hipMemcpyHostToDevice, stream) != hipSuccess ||
            hipEventRecord(done[slot], stream) != hipSuccess) { rc = fail(SVT_ERR_HIP, "staged hipMemcpyAsync"); break; }
        off += len;
    }
    (void)hipStreamSynchronize(stream);   // the ring is reusable once the last piece has left
    for (int i = 0; i < StagingRing::kSlots; ++i)
        if (done[i]) (void)hipEventDestroy(done[i]);
    return rc;
}

// Device -> host through the same pinned ring (results: 128 B per unit).
inline int d2h_staged(void* dst, const void* src, uint64_t bytes, hipStream_t stream)
{
    if (bytes < (16ull << 20)) {
        if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SVT_OK;
    }
    std::lock_guard<std::mutex> guard(g_ring.lock);
    SVT_TRY(g_ring.ensure());
    const unsigned nt = std::min(host_threads(), 6u);
    // piece k is copied out of its slot while piece k + 1 is on the wire
    uint64_t off = 0, prev_off = 0, prev_len = 0;
    int slot = 0, prev_slot = -1;
    while (off < bytes || prev_slot >= 0) {
        uint64_t len = 0;
        if (off < bytes) {
            len = std::min(StagingRing::kPiece, bytes - off);
            HIP_TRY(hipMemcpyAsync(g_ring.buf[slot], static_cast<const char*>(src) + off, len, hipMemcpyDeviceToHost, stream));
        }
        if (prev_slot >= 0) {
            const char* p0 = static_cast<const char*>(g_ring.buf[prev_slot]);
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
