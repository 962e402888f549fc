// Which engine moves a device-to-host copy, by kind of host memory?  (run under rocprofv3 --kernel-trace --memory-copy-trace)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
    const size_t n = 1ull << 30;
    void *d; CK(hipMalloc(&d, n)); CK(hipMemset(d, 1, n));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct { const char *name; unsigned flags; } kinds[] = {{"default", hipHostMallocDefault}, {"noncoherent", hipHostMallocNonCoherent},
        {"portable|mapped", hipHostMallocPortable | hipHostMallocMapped}, {"coherent", hipHostMallocCoherent}, {"numa", hipHostMallocNumaUser}};
    for (auto &k : kinds) {
        void *h; if (hipHostMalloc(&h, n, k.flags) != hipSuccess) { printf("%s: alloc failed\n", k.name); (void)hipGetLastError(); continue; }
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
            auto t2 = std::chrono::steady_clock::now();
            printf("%-16s D2H %.1f GB/s  H2D %.1f GB/s\n", k.name, n / 1e9 / std::chrono::duration<double>(t1 - t0).count(), n / 1e9 / std::chrono::duration<double>(t2 - t1).count());
        }
        CK(hipHostFree(h));
    }
    void *p = malloc(n); CK(hipHostRegister(p, n, hipHostRegisterDefault));
    auto t0 = std::chrono::steady_clock::now();
    CK(hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    auto t1 = std::chrono::steady_clock::now();
    printf("registered       D2H %.1f GB/s\n", n / 1e9 / std::chrono::duration<double>(t1 - t0).count());
    return 0;
}
