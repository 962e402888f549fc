#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void spin(unsigned long long* out, int iters) {
    extern __shared__ unsigned lds[];
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned v = threadIdx.x;
    for (int i = 0; i < iters; i++) { lds[threadIdx.x] = v; v = lds[(threadIdx.x + 1) & 63] * 3 + 1; }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t0; out[blockIdx.x * 2 + 1] = t1 + (v == 12345); }
}
int main() {
    for (int ldsb : {4096, 16384, 20288, 20480, 24576, 32768, 40960, 65536}) {
        int nb = 0;
        hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin, 64, ldsb);
        const int grid = 8192;
        unsigned long long* d; hipMalloc(&d, grid * 16);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(spin, dim3(grid), dim3(64), ldsb, 0, d, 1000);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(spin, dim3(grid), dim3(64), ldsb, 0, d, 20000);
        hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        // single block time
        hipEventRecord(a);
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), ldsb, 0, d, 20000);
        hipEventRecord(b); hipDeviceSynchronize();
        float ms1; hipEventElapsedTime(&ms1, a, b);
        printf("lds %6d B: occupancy API %2d blocks/CU; 8192 blocks %.2f ms, 256 blocks %.2f ms -> rounds %.1f -> resident/CU ~ %.1f\n",
               ldsb, nb, ms, ms1, ms / ms1, 8192.0 / 256 / (ms / ms1));
        hipFree(d);
    }
    return 0;
}
