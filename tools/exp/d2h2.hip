// Does a device-to-host copy stay on the copy engines while a host-to-device copy runs, and with odd offsets / sizes?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
    const size_t n = 1ull << 30;
    char *d, *d2; CK(hipMalloc((void **)&d, n)); CK(hipMalloc((void **)&d2, n)); CK(hipMemset(d, 1, n));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    char *h, *h2; CK(hipHostMalloc((void **)&h, n, 0)); CK(hipHostMalloc((void **)&h2, n, 0));
    // 1: both directions at once, three rounds
    for (int r = 0; r < 3; r++) { CK(hipMemcpyAsync(d2, h2, n, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s2)); }
    CK(hipDeviceSynchronize());
    // 2: alone, odd offset and size
    CK(hipMemcpyAsync(h + 12 * 1000003, d + 256, n / 2 + 12 * 777, hipMemcpyDeviceToHost, s2)); CK(hipDeviceSynchronize());
    // 3: D2H submitted while 4 H2D are queued on the other stream
    for (int r = 0; r < 4; r++) CK(hipMemcpyAsync(d2, h2, n / 4, hipMemcpyHostToDevice, s1));
    CK(hipMemcpyAsync(h, d, n / 2, hipMemcpyDeviceToHost, s2));
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
