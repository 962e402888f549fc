// VALU issue-rate microbenchmark (inline asm, nothing for the compiler to fold) for the integer
// ops the fill kernel uses, gfx950.  8 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPCASE(N, TXT) if (OP == N) asm volatile(TXT : "+v"(x[i]) : "v"(b), "v"(c));
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned a0, unsigned b0) {
    unsigned x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * (i + 1) + a0;
    unsigned b = b0 + threadIdx.x, c = b0 * 3 + 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                OPCASE(0, "v_add_u32 %0, %0, %1")
                OPCASE(1, "v_min3_i32 %0, %0, %1, %2")
                OPCASE(2, "v_perm_b32 %0, %0, %1, %2")
                OPCASE(3, "v_and_or_b32 %0, %0, %1, %2")
                OPCASE(4, "v_lshl_add_u32 %0, %0, 5, %1")
                OPCASE(5, "v_bfi_b32 %0, %2, %0, %1")
                OPCASE(6, "v_lshl_or_b32 %0, %0, 3, %1")
                OPCASE(7, "v_alignbit_b32 %0, %0, %1, 2")
                OPCASE(8, "v_and_b32 %0, %0, %1")
                OPCASE(9, "v_bfe_u32 %0, %0, 3, 2")
                OPCASE(10, "v_or3_b32 %0, %0, %1, %2")
                OPCASE(11, "v_lshlrev_b32 %0, 3, %0")
                OPCASE(12, "v_cndmask_b32 %0, %0, %1, vcc")
                OPCASE(13, "v_min_i32 %0, %0, %1")
                OPCASE(14, "v_add3_u32 %0, %0, %1, %2")
                OPCASE(15, "v_mov_b32 %0, %1")
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s ^= x[i];
    if (s == 0x12345) out[0] = s;
}
template <int OP> void run(const char* name, unsigned* d, int wavesPerSimd) {
    const int iters = 20000;
    const int blocks = 256 * wavesPerSimd;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, 1u, 2u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u, 2u);
    (void)hipEventRecord(b); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double winst = (double)blocks * 4 * iters * 64;
    printf("%-16s %d waves/SIMD: %7.2f ms  %6.2f T lane-ops/s  %.2f ns per wave-instr per SIMD\n", name, wavesPerSimd, ms,
           winst * 64 / ms / 1e9, ms * 1e6 / (winst / 1024));
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 1024);
    for (int w : {1, 2, 4}) {
        run<0>("v_add_u32", d, w); run<1>("v_min3_i32", d, w); run<2>("v_perm_b32", d, w); run<3>("v_and_or_b32", d, w);
        run<4>("v_lshl_add_u32", d, w); run<5>("v_bfi_b32", d, w); run<6>("v_lshl_or_b32", d, w); run<7>("v_alignbit_b32", d, w);
        run<8>("v_and_b32", d, w); run<9>("v_bfe_u32", d, w); run<10>("v_or3_b32", d, w); run<11>("v_lshlrev_b32", d, w);
        run<12>("v_cndmask_b32", d, w); run<13>("v_min_i32", d, w); run<14>("v_add3_u32", d, w); run<15>("v_mov_b32", d, w);
    }
    return 0;
}
