// Probe: stream 16-byte-per-lane rows through a per-wave LDS ring filled by global_load_lds
// (no VGPR staging) vs. plain register loads; prints GB/s and a checksum of both.
// build: hipcc --offload-arch=gfx950 -O3 tools/glds_probe.hip -o tools/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WAVES, int D>
__global__ __launch_bounds__(WAVES * 64) void ring_kernel(const uint4* __restrict__ g, unsigned long long* out, int rows)
{
    __shared__ __attribute__((aligned(16))) uint4 ring_all[WAVES * D * 64];
    const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    uint4* ring = ring_all + wave * D * 64;
    const uint4* p = g + ((size_t)blockIdx.x * WAVES + wave) * rows * 64 + lane;
#pragma unroll
    for (int k = 0; k < D; ++k)
        __builtin_amdgcn_global_load_lds(p + k * 64, (lds_ptr)(ring + k * 64), 16, 0, 2);
    unsigned long long acc = 0;
    const uint32_t ring_addr = (uint32_t)(uintptr_t)(lds_ptr)(ring + lane);
    for (int j = 0; j < rows; ++j) {
        uint4 w;
        const uint32_t slot = (uint32_t)j % D;
        asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(w) : "v"(ring_addr + slot * 1024), "n"(D - 1) : "memory");
        __builtin_amdgcn_global_load_lds(p + (size_t)(j + D) * 64, (lds_ptr)(ring + slot * 64), 16, 0, 2);
        acc += (unsigned long long)w.x * 3 + w.y + ((unsigned long long)w.z << 7) + w.w * (unsigned long long)(j + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int WAVES, int G>
__global__ __launch_bounds__(WAVES * 64) void reg_kernel(const uint4* __restrict__ g, unsigned long long* out, int rows)
{
    const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const uint4* p = g + ((size_t)blockIdx.x * WAVES + wave) * rows * 64 + lane;
    unsigned long long acc = 0;
    for (int j = 0; j < rows; j += G) {
        uint4 w[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const uint4* q = p + (size_t)(j + k) * 64;
            w[k].x = __builtin_nontemporal_load(&q->x); w[k].y = __builtin_nontemporal_load(&q->y);
            w[k].z = __builtin_nontemporal_load(&q->z); w[k].w = __builtin_nontemporal_load(&q->w);
        }
#pragma unroll
        for (int k = 0; k < G; ++k)
            acc += (unsigned long long)w[k].x * 3 + w[k].y + ((unsigned long long)w[k].z << 7) + w[k].w * (unsigned long long)(j + k + 1);
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F> static float time_it(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

static unsigned long long checksum(unsigned long long* d, size_t n)
{
    std::vector<unsigned long long> h(n); CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    unsigned long long s = 0; for (auto v : h) s = s * 1000003ULL + v; return s;
}

template <int WAVES, int D> static void run_ring(const uint4* g, unsigned long long* out, size_t waves_total, int rows, double bytes)
{
    const int grid = (int)(waves_total / WAVES);
    float ms = time_it([&] { hipLaunchKernelGGL((ring_kernel<WAVES, D>), dim3(grid), dim3(WAVES * 64), 0, 0, g, out, rows); }, 10);
    printf("ring  waves/wg %2d depth %2d: %.3f ms  %.0f GB/s  sum %016llx\n", WAVES, D, ms, bytes / ms / 1e6, checksum(out, waves_total * 64));
}
template <int WAVES, int G> static void run_reg(const uint4* g, unsigned long long* out, size_t waves_total, int rows, double bytes)
{
    const int grid = (int)(waves_total / WAVES);
    float ms = time_it([&] { hipLaunchKernelGGL((reg_kernel<WAVES, G>), dim3(grid), dim3(WAVES * 64), 0, 0, g, out, rows); }, 10);
    printf("reg   waves/wg %2d group %2d: %.3f ms  %.0f GB/s  sum %016llx\n", WAVES, G, ms, bytes / ms / 1e6, checksum(out, waves_total * 64));
}

int main(int argc, char** argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 64;           // rows of 64 x 16 B per wave
    const size_t waves_total = 16384;                          // 16384 x rows KB
    const size_t n = waves_total * (size_t)(rows + 32) * 64;   // + pad rows for the look-ahead
    uint4* g; CK(hipMalloc(&g, n * 16));
    std::vector<uint4> h(n);
    unsigned long long x = 88172645463325252ULL;
    for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)(x * 7), (uint32_t)(x >> 13)); }
    CK(hipMemcpy(g, h.data(), n * 16, hipMemcpyHostToDevice));
    unsigned long long* out; CK(hipMalloc(&out, waves_total * 64 * 8));
    const double bytes = (double)waves_total * rows * 1024;
    printf("rows/wave %d, %.1f MB\n", rows, bytes / 1e6);
    run_reg<4, 2>(g, out, waves_total, rows, bytes);
    run_reg<4, 4>(g, out, waves_total, rows, bytes);
    run_reg<4, 8>(g, out, waves_total, rows, bytes);
    run_ring<4, 8>(g, out, waves_total, rows, bytes);
    run_ring<8, 8>(g, out, waves_total, rows, bytes);
    run_ring<16, 8>(g, out, waves_total, rows, bytes);
    run_ring<16, 4>(g, out, waves_total, rows, bytes);
    run_ring<8, 16>(g, out, waves_total, rows, bytes);
    run_ring<4, 16>(g, out, waves_total, rows, bytes);
    return 0;
}
