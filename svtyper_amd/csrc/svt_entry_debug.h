// svt_entry_debug.h -- part of the single translation unit svtyper_hip.hip (included there, in order; not a stand-alone header):
// measurement hooks of the tools (not in include/svtyper_hip.h).

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

