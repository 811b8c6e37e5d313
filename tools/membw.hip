// Streaming-read ceiling probe for this box: reads N bytes with 16 B/lane loads in the same
// row-strided pattern as svt_genotype_kernel (wave reads 1 KiB rows), xors into a dummy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NT, int G>
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ p, uint64_t rows_per_wave, u32x4* out)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u32x4* q = p + wave * rows_per_wave * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t j = 0; j + G <= rows_per_wave; j += G) {
        u32x4 v[G];
#pragma unroll
        for (int k = 0; k < G; ++k) v[k] = NT ? __builtin_nontemporal_load(q + (j + k) * 64) : q[(j + k) * 64];
#pragma unroll
        for (int k = 0; k < G; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[wave * 64 + lane] = acc;
}
template <int NT, int G>
void run(const u32x4* d, uint64_t bytes, uint64_t rows, u32x4* out, const char* name)
{
    const uint64_t waves = bytes / (rows * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rd<NT, G>), dim3(waves / 4), dim3(256), 0, 0, d, rows, out);
    hipEventRecord(a);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((rd<NT, G>), dim3(waves / 4), dim3(256), 0, 0, d, rows, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s rows/wave=%4llu  %.3f ms  %.1f GB/s\n", name, (unsigned long long)rows, ms / it, bytes / (ms / it * 1e-3) / 1e9);
}
int main()
{
    const uint64_t bytes = 1600ull << 20;
    u32x4 *d, *out;
    hipMalloc(&d, bytes); hipMalloc(&out, 64 << 20);
    hipMemset(d, 1, bytes);
    for (uint64_t rows : {100ull, 400ull, 1600ull}) {
        run<0, 4>(d, bytes, rows, out, "plain loads, group 4");
        run<1, 4>(d, bytes, rows, out, "nt loads, group 4");
        run<1, 8>(d, bytes, rows, out, "nt loads, group 8");
    }
    return 0;
}
