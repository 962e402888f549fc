"""Host<->device copy rates on this box: pageable and pinned, each direction, and both directions at once."""
import time
import torch

def rate(fn, nbytes, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e9

n = 1 << 30
d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
d2 = torch.empty(n, dtype=torch.uint8, device="cuda:0")
hp = torch.empty(n, dtype=torch.uint8).pin_memory()
hp2 = torch.empty(n, dtype=torch.uint8).pin_memory()
hg = torch.empty(n, dtype=torch.uint8)
hg.fill_(1)
print("H2D pageable GB/s", round(rate(lambda: d.copy_(hg), n), 1))
print("D2H pageable GB/s", round(rate(lambda: hg.copy_(d), n), 1))
print("H2D pinned   GB/s", round(rate(lambda: d.copy_(hp, non_blocking=True), n), 1))
print("D2H pinned   GB/s", round(rate(lambda: hp.copy_(d, non_blocking=True), n), 1))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1):
        d.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2):
        hp2.copy_(d2, non_blocking=True)
print("both ways pinned, GB/s each way", round(rate(both, n), 1))
t0 = time.perf_counter(); x = torch.empty(n, dtype=torch.uint8).pin_memory(); print("pin 1 GiB s", round(time.perf_counter() - t0, 3))
import numpy as np
a = np.ones(n, np.uint8); b = np.empty(n, np.uint8)
t0 = time.perf_counter(); b[:] = a; print("host memcpy 1 thread GB/s", round(n / (time.perf_counter() - t0) / 1e9, 1))
