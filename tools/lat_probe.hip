// Probe: what the instructions of the record consumer cost ONE wave that has its SIMD to itself (launches of less than one
// round of resident workgroups run that way): dependent / independent v_add_f64, LDS reads (dependent chain; 64 random
// addresses in a 2 KB table, b64 / u16), DPP moves, ds_bpermute, a dependent global load.  Prints shader cycles per
// instruction (s_memtime) and the shader clock (s_memtime against the 100 MHz s_memrealtime).
//   build: hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o tools/lat_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Out { uint64_t cyc[16]; uint64_t real[16]; double sink; };

__global__ void probe(Out* out, const uint32_t* chase, const uint32_t* rnd, int iters)
{
    __shared__ double tab[256 + 1024];
    __shared__ uint32_t next[1024];
    for (int i = threadIdx.x; i < 256 + 1024; i += blockDim.x) tab[i] = 1.0 + i * 1e-9;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) next[i] = (uint32_t)((i * 389 + 77) & 1023) * 4u;
    __syncthreads();
    double acc = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0;
    const uint32_t lane = threadIdx.x & 63;
    uint64_t t0, t1, r0, r1;
    int slot = 0;
#define BEGIN() r0 = wall_clock64(); t0 = clock64();
#define END() t1 = clock64(); r1 = wall_clock64(); if (threadIdx.x == 0 && blockIdx.x == 0) { out->cyc[slot] = t1 - t0; out->real[slot] = r1 - r0; } ++slot;
    // 0: dependent f64 adds
    double x = 1.0 + lane * 1e-12;
    BEGIN();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(x));
    }
    END();
    // 1: six independent f64 add chains
    BEGIN();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(x));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a1) : "v"(x));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(x));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a3) : "v"(x));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a4) : "v"(x));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a5) : "v"(x));
        }
    }
    END();
    // 2: dependent ds_read_b32 chain (LDS latency)
    uint32_t p = lane * 4u;
    const uint32_t next_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)next;
    BEGIN();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, %2" : "+v"(p) : "v"(p), "v"(0u));
        p = (p & 4095u) + next_base - next_base;
    }
    END();
    // 3: 16 independent ds_read_b64 with random addresses in a 2 KB table, then one wait
    const uint32_t tab_base = (uint32_t)(size_t)(__attribute__((address_space(3))) double*)tab;
    uint32_t ra[16];
    for (int j = 0; j < 16; ++j) ra[j] = tab_base + (rnd[(lane * 16 + j) & 4095] & 255u) * 8u;
    BEGIN();
    for (int i = 0; i < iters; ++i) {
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("ds_read_b64 %0, %1" : "=v"(v[j]) : "v"(ra[j]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(v[j]));
        acc += v[0];
    }
    END();
    // 4: the same where all lanes read few distinct addresses (MAPQs cluster): 4 distinct
    for (int j = 0; j < 16; ++j) ra[j] = tab_base + (rnd[(lane * 16 + j) & 4095] & 3u) * 8u * 60u;
    BEGIN();
    for (int i = 0; i < iters; ++i) {
        double v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("ds_read_b64 %0, %1" : "=v"(v[j]) : "v"(ra[j]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(v[j]));
        acc += v[0];
    }
    END();
    // 5: 16 independent ds_read_u16 random in 2 KB
    for (int j = 0; j < 16; ++j) ra[j] = tab_base + (rnd[(lane * 16 + j) & 4095] & 1023u) * 2u;
    BEGIN();
    for (int i = 0; i < iters; ++i) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("ds_read_u16 %0, %1" : "=v"(v[j]) : "v"(ra[j]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(v[j]));
        p += v[0];
    }
    END();
    // 6: DPP moves (row_shr:1), dependent
    uint32_t q = lane;
    BEGIN();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(q));
    }
    END();
    // 7: ds_bpermute, dependent
    uint32_t bp = ((lane + 1) & 63) * 4u;
    BEGIN();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(q) : "v"(bp));
    }
    END();
    // 8: dependent global loads (a chase through 64 MB: HBM / MALL latency)
    uint32_t g = lane * 1024u + blockIdx.x * 77u;
    BEGIN();
    for (int i = 0; i < iters / 4 + 1; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) g = __builtin_nontemporal_load(chase + (g & ((16u << 20) - 1u)));
    }
    END();
    // 9: v_readlane + v_add_f64 with an SGPR pair operand (a broadcast addend), dependent adds
    BEGIN();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t lo, hi;
            asm volatile("v_readlane_b32 %0, %2, %4\n\tv_readlane_b32 %1, %3, %4" : "=s"(lo), "=s"(hi) : "v"(__double2loint(x)), "v"(__double2hiint(x)), "n"(5));
            const double s = __hiloint2double((int)hi, (int)lo);
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "s"(s));
        }
    }
    END();
    if (acc + a1 + a2 + a3 + a4 + a5 == 12345.678 || p == 77u || q == 0xdeadu || g == 0xbeefu) out->sink = acc;
}

int main(int argc, char** argv)
{
    const int iters = 2000;
    Out* d_out;
    CHECK(hipMalloc(&d_out, sizeof(Out)));
    std::vector<uint32_t> chase(16u << 20), rnd(4096);
    uint64_t s = 88172645463325252ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (auto& c : chase) c = next();
    for (auto& r : rnd) r = next();
    uint32_t *d_chase, *d_rnd;
    CHECK(hipMalloc(&d_chase, chase.size() * 4));
    CHECK(hipMalloc(&d_rnd, rnd.size() * 4));
    CHECK(hipMemcpy(d_chase, chase.data(), chase.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rnd, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice));
    const char* names[] = {"dependent v_add_f64", "6 independent v_add_f64 chains (per 6)", "dependent ds_read_b32 (+ add)", "16 x ds_read_b64 random/2KB + wait (per 16)",
                           "16 x ds_read_b64 4 distinct (per 16)", "16 x ds_read_u16 random (per 16)", "dependent v_mov_dpp row_shr (+ s_nop 1)", "dependent ds_bpermute",
                           "dependent global nt load (64 MB chase)", "2 v_readlane + v_add_f64 sgpr (dependent)"};
    const double per[] = {16, 16, 16, 1, 1, 1, 16, 16, 0, 16};
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int blocks = cfg == 0 ? 1 : cfg == 1 ? 256 : 1024, threads = cfg == 2 ? 256 : 64;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d_out, d_chase, d_rnd, iters);
            CHECK(hipDeviceSynchronize());
        }
        Out o;
        CHECK(hipMemcpy(&o, d_out, sizeof(o), hipMemcpyDeviceToHost));
        std::printf("== %d workgroups x %d threads\n", blocks, threads);
        for (int k = 0; k < 10; ++k) {
            const double n = k == 8 ? (iters / 4 + 1) * 4.0 : iters * per[k];
            std::printf("  %-46s %8.1f cycles   %8.1f ns   (clock %.0f MHz)\n", names[k], o.cyc[k] / n, o.real[k] * 10.0 / n, o.cyc[k] / (o.real[k] * 10.0) * 1000.0);
        }
    }
    return 0;
}
