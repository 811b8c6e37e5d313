// Probe: the access pattern of svt_stream_kernel without its arithmetic -- one lane per unit, every step each unit
// fetches its next LINES x 128 bytes by LDS-DMA into a per-wave ring row, the lane reads the row into VGPRs, the
// next fetch is issued, the row is "consumed" (xor + a dependent VALU chain of `spin` steps per 8 records).
//   LINES  contiguous 128-byte lines per unit and request (1 = what the kernel does; 2, 4: fewer DRAM row activations)
//   DEPTH  ring stages per wave (requests in flight per unit)
//   WAVES  waves per workgroup; `pad` bytes of unused LDS per workgroup set the workgroups per CU
// build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// WR: 0 no result records, 1 plain / 2 nt full 128-byte lines for consecutive units, 3 plain / 4 nt for units scattered over
// the workgroup's range (what a length-sorted tile writes), 5 / 6: sc1 / sc0 sc1 scattered
template <int LINES, int DEPTH, int WAVES, int AUX, int WR>
__global__ __launch_bounds__(WAVES * 64) void gather_kernel(const char* __restrict__ g, uint32_t stride, uint32_t steps, uint32_t spin,
                                                            uint32_t ring_off, unsigned long long* out, char* res)
{
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    constexpr uint32_t kRow = LINES * 128u, kStage = 64u * kRow, kInstr = kStage / 1024u;   // DMA instructions per stage
    constexpr uint32_t LPU = 8u * LINES, UPI = 64u / LPU;                                  // lanes per unit, units per instruction
    const uint32_t lane = threadIdx.x % 64u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64u));
    unsigned char* ring = smem + ring_off + wave * (DEPTH * kStage);
    const uint64_t unit0 = ((uint64_t)blockIdx.x * WAVES + wave) * 64u;
    const char* src0 = g + (unit0 + lane / LPU) * (uint64_t)stride + (lane % LPU) * 16u;   // instruction 0 of step 0
    const uint32_t ring_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const uint32_t row_addr = ring_addr + lane * kRow;
    const uint32_t swz = LINES == 1 ? ((lane >> 1) & 7u) : (lane & (LPU - 1u));

    auto fetch = [&](const uint32_t s) {
        unsigned char* st = ring + (s % DEPTH) * kStage;
#pragma unroll
        for (uint32_t i = 0; i < kInstr; ++i)
            __builtin_amdgcn_global_load_lds(src0 + (uint64_t)i * UPI * stride + (uint64_t)s * kRow, (lds_ptr)(st + i * 1024u), 16, 0, AUX);
    };
#pragma unroll
    for (uint32_t s = 0; s < DEPTH; ++s)
        if (s < steps) fetch(s);
    u32x4 acc = {0, 0, 0, 0};
    float chain = (float)lane;
    for (uint32_t s = 0; s < steps; ++s) {
        u32x4 w[LINES * 8];
        const uint32_t base = row_addr + (s % DEPTH) * kStage;
        // the oldest stage has landed when at most (DEPTH - 1) stages' worth of instructions are outstanding
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (s + DEPTH <= steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * (int)kInstr > 63 ? 63 : (DEPTH - 1) * (int)kInstr) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (uint32_t j = 0; j < LINES * 8; ++j)
            asm volatile("ds_read_b128 %0, %1" : "=v"(w[j]) : "v"(base + ((j ^ swz) << 4)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s + DEPTH < steps) fetch(s + DEPTH);
#pragma unroll
        for (uint32_t j = 0; j < LINES * 8; ++j) acc ^= w[j];
        for (uint32_t k = 0; k < spin * LINES; ++k) chain = chain * 1.0001f + 0.5f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (WR) {
        const uint32_t o = lane >> 3, rr = lane & 7u;
        const uint64_t wg0 = (uint64_t)blockIdx.x * WAVES * 64u;
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
            uint32_t local = wave * 64u + 8u * i + o;                       // unit inside the workgroup
            if (WR >= 3) local = (local * 37u + 11u) % (WAVES * 64u);       // a permutation: 37 is odd
            char* dst = res + (wg0 + local) * 128u + rr * 16u;
            const u32x4 v = {acc.x + i, acc.y, acc.z, (uint32_t)chain};
            if (WR == 1 || WR == 3) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
            else if (WR == 2 || WR == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
            else if (WR == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (unsigned long long)chain;
}

template <typename F> static float time_it(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int t = 0; t < 3; ++t) {
        CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms / reps < best) best = ms / reps;
    }
    return best;
}

static const char* g_buf; static char* g_res; static unsigned long long* g_out; static size_t g_units; static uint32_t g_stride, g_lines_per_unit;

template <int LINES, int DEPTH, int WAVES, int AUX = 2, int WR = 0>
static void run(uint32_t lds_per_wg_kb, uint32_t spin)
{
    const uint32_t rings = WAVES * DEPTH * 64u * LINES * 128u;
    uint32_t lds = lds_per_wg_kb * 1024u;
    if (lds < rings) lds = rings;
    const uint32_t ring_off = (lds - rings) & ~127u;
    CK(hipFuncSetAttribute((const void*)gather_kernel<LINES, DEPTH, WAVES, AUX, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = (int)(g_units / (WAVES * 64));
    const uint32_t steps = g_lines_per_unit / LINES;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_kernel<LINES, DEPTH, WAVES, AUX, WR>, WAVES * 64, lds));
    float ms = time_it([&] { hipLaunchKernelGGL((gather_kernel<LINES, DEPTH, WAVES, AUX, WR>), dim3(grid), dim3(WAVES * 64), lds, 0, g_buf, g_stride, steps, spin, ring_off, g_out, g_res); }, 5);
    const double bytes = (double)g_units * g_lines_per_unit * 128.0;
    printf("wr %d lines %d depth %d waves/wg %2d aux %d lds %3u KB wg/cu %d (waves/cu %2d, %3u KB in flight/cu) spin %4u: %.4f ms  %6.0f GB/s\n", WR, LINES, DEPTH, WAVES, AUX,
           lds / 1024u, occ, occ * WAVES, occ * WAVES * DEPTH * 64u * LINES * 128u / 1024u, spin, ms, bytes / ms / 1e6);
    fflush(stdout);
}

// a buffer of `bytes` built from physical chunks of `chunk` bytes mapped into one virtual range (hipMemCreate / hipMemMap)
static char* vmm_alloc(size_t bytes, size_t chunk)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t total = (bytes + chunk - 1) / chunk * chunk;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap(static_cast<char*>(va) + off, chunk, 0, h, 0));
        CK(hipMemRelease(h));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    return static_cast<char*>(va);
}

int main(int argc, char** argv)
{
    // does the physical chunking of the buffer decide the level?  plain hipMalloc vs virtual-memory mappings of chunks of several sizes
    g_units = (size_t)1 << 20;
    g_lines_per_unit = 12;
    g_stride = 1664u;
    const size_t bytes = g_units * (size_t)g_stride + (1 << 20);
    CK(hipMalloc(&g_out, g_units * 8));
    CK(hipMalloc(&g_res, g_units * 128));
    char* res_plain = g_res;
    char* res_vmm = vmm_alloc(g_units * 128, (size_t)256 << 20);
    for (int rep = 0; rep < 4; ++rep) {
        { char* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); g_buf = p; g_res = res_plain; printf("hipMalloc data, hipMalloc results #%d: ", rep); run<1, 1, 4, 2, 4>(49, 150); }
        { char* p = vmm_alloc(bytes, (size_t)2048 << 20); CK(hipMemset(p, 1, bytes)); g_buf = p; g_res = res_plain; printf("one 2 GB chunk data, hipMalloc results #%d: ", rep); run<1, 1, 4, 2, 4>(49, 150);
          g_res = res_vmm; printf("one 2 GB chunk data, mapped results #%d: ", rep); run<1, 1, 4, 2, 4>(49, 150); }
        { char* p = vmm_alloc(bytes, (size_t)256 << 20); CK(hipMemset(p, 1, bytes)); g_buf = p; g_res = res_plain; printf("256 MB chunks data, hipMalloc results #%d: ", rep); run<1, 1, 4, 2, 4>(49, 150); }
        { char* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); g_buf = p; g_res = res_vmm; printf("hipMalloc data, mapped results #%d: ", rep); run<1, 1, 4, 2, 4>(49, 150); }
        { char* r; CK(hipMalloc(&r, g_units * 128)); char* p = vmm_alloc(bytes, (size_t)2048 << 20); CK(hipMemset(p, 1, bytes)); g_buf = p; g_res = r; printf("one 2 GB chunk data, a new hipMalloc for results #%d: ", rep); run<1, 1, 4, 2, 4>(49, 150); }
    }
    return 0;
}
