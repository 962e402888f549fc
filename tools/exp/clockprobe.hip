// One wave that measures the core clock while other work runs: each sample is a fixed chain of dependent integer
// operations timed with the constant 100 MHz counter.  probe_start launches it on a stream of its own, probe_finish
// waits for it and copies the samples (start tick, ticks taken) out.  The loops are finite: the kernel always ends.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(64) void k_probe(uint64_t *out, int samples, int inner)
{
    uint32_t x = threadIdx.x + 1u;
    for (int i = 0; i < samples; i++) {
        const uint64_t t0 = wall_clock64();
        for (int k = 0; k < inner; k++) {
#pragma unroll
            for (int u = 0; u < 16; u++) x = x * 3u + 1u;         // dependent chain
        }
        const uint64_t t1 = wall_clock64();
        if (threadIdx.x == 0) { out[2 * i] = t0; out[2 * i + 1] = t1 - t0 + (x == 0x12345u ? 1u : 0u); }
    }
}

static hipStream_t g_s;
static uint64_t *g_out;
static int g_samples;

extern "C" int probe_start(int samples, int inner)
{
    if (hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking) != hipSuccess) return 1;
    if (hipMalloc((void **)&g_out, (size_t)samples * 16) != hipSuccess) return 2;
    g_samples = samples;
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, g_s, g_out, samples, inner);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int probe_finish(uint64_t *host)
{
    if (hipStreamSynchronize(g_s) != hipSuccess) return 1;
    if (hipMemcpy(host, g_out, (size_t)g_samples * 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    (void)hipFree(g_out);
    (void)hipStreamDestroy(g_s);
    return 0;
}
