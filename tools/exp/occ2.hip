#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void spin(unsigned* out, int iters) {
    extern __shared__ unsigned lds[];
    unsigned v = threadIdx.x + blockIdx.x, w = v * 7;
    lds[threadIdx.x] = v;
    for (int i = 0; i < iters; i++) { v = v * 1664525u + 1013904223u; w ^= v >> 3; w += v; }
    if (v == 12345 && w == 7) out[0] = lds[(threadIdx.x + 1) & 63];
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 1024);
    (void)hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int ldsb : {1024, 20288, 36000}) {
        for (int grid : {256, 1024, 1536, 2048, 3072, 4096, 8192}) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            hipLaunchKernelGGL(spin, dim3(grid), dim3(64), ldsb, 0, d, 1000);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(grid), dim3(64), ldsb, 0, d, 2000000);
            (void)hipEventRecord(b); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf("lds %6d grid %5d (%4.1f waves/CU): %.2f ms\n", ldsb, grid, grid / 256.0, ms);
        }
    }
    return 0;
}
