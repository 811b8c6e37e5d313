// H2D ceiling probe: pinned -> device rate, multi-threaded host memcpy rate, pageable hipMemcpy rate.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = 1600ull << 20, piece = 32ull << 20;
    char* src = (char*)malloc(bytes); memset(src, 1, bytes);
    void *d, *pin; hipMalloc(&d, bytes); hipHostMalloc(&pin, piece, 0);
    double t = now(); hipMemcpy(d, src, bytes, hipMemcpyHostToDevice); printf("pageable hipMemcpy      %.1f GB/s\n", bytes / (now() - t) / 1e9);
    t = now(); hipMemcpy(d, src, bytes, hipMemcpyHostToDevice); printf("pageable hipMemcpy (2nd) %.1f GB/s\n", bytes / (now() - t) / 1e9);
    t = now(); for (size_t o = 0; o < bytes; o += piece) hipMemcpyAsync((char*)d + o, pin, piece, hipMemcpyHostToDevice, 0); hipDeviceSynchronize();
    printf("pinned -> device         %.1f GB/s\n", bytes / (now() - t) / 1e9);
    for (unsigned nt : {1u, 4u, 8u, 16u, 32u}) {
        t = now();
        for (size_t o = 0; o < bytes; o += piece) {
            std::vector<std::thread> th; size_t part = piece / nt;
            for (unsigned k = 0; k < nt; ++k) th.emplace_back([=]() { memcpy((char*)pin + k * part, src + o + k * part, part); });
            for (auto& x : th) x.join();
        }
        printf("host memcpy %2u threads   %.1f GB/s\n", nt, bytes / (now() - t) / 1e9);
    }
    t = now(); hipHostRegister(src, bytes, 0); double tr = now() - t;
    t = now(); hipMemcpy(d, src, bytes, hipMemcpyHostToDevice); printf("hipHostRegister %.1f ms, then copy %.1f GB/s\n", tr * 1e3, bytes / (now() - t) / 1e9);
    return 0;
}
