// How does the runtime deal streams to hardware queues?  Streams are created in order 0..7, first used in the order
// given on the command line; the kernel trace shows the queue of each launch (grid size = stream index + 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_probe(int *p) { if (p) p[0] = 1; }
int main(int argc, char **argv)
{
    hipStream_t s[8];
    for (int i = 0; i < 8; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    for (int a = 1; a < argc; a++) {
        int i = atoi(argv[a]);
        hipLaunchKernelGGL(k_probe, dim3(i + 1), dim3(64), 0, s[i], nullptr);
        hipStreamSynchronize(s[i]);
    }
    hipDeviceSynchronize();
    return 0;
}
