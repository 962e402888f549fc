"""The core clock while one-part batches run: a one-wave probe kernel (tools/exp/clockprobe.hip, built by hand with hipcc
into tools/exp/libclockprobe.so) times a fixed chain of dependent operations against the constant 100 MHz counter all
the while; the batch is the Local-end-rule leg of the bench line (122 880 pairs of 10 kb).  Prints per step the tail the
library measured and, per 4 ms of the probe's time line, the chain's duration (it is inversely proportional to the clock).
usage: tail_probe.py [steps = 6]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # sets GPU_MAX_HW_QUEUES before the runtime starts
import numpy as np
import torch
from blasr_amd import api, synth, _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = 122880
cfg = dict(synth.PRESETS["c2_10kb_linear"])
ref = synth.reference(cfg["ref_len"], synth.REF_SEED, cfg.get("n_frac", 0.0))
host = synth.generate(n, ref=ref, **{a: b for a, b in cfg.items() if a != "ref"})
d = api.to_device(host, "cuda:0")
cap = int(host["q_len"].astype(np.int64).sum() // 4) + n
out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
out_blocks = torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0")
ctx = api.Context(0, timing=True)
ctx.set_scoring(api.make_scoring(band=cfg["band"], align_type=_lib.ALIGN_LOCAL))
ctx.refine_device(d, out_pairs, out_blocks)
torch.cuda.synchronize()
P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libclockprobe.so"))
samples, inner = 160000, 256                   # a sample is about 4 us at full clock
assert P.probe_start(samples, inner) == 0
time.sleep(0.02)
tails = []
for _ in range(steps):
    ctx.refine_device(d, out_pairs, out_blocks)
    torch.cuda.synchronize()
    tm = ctx.timing()
    tails.append((round(tm["fill_busy_ms"], 1), round(tm["traceback_ms"], 1)))
    time.sleep(0.03)
buf = np.zeros(2 * samples, np.uint64)
assert P.probe_finish(buf.ctypes.data_as(C.c_void_p)) == 0
ctx.close()
print("fill / tail per step (ms):", tails)
t = (buf[0::2] - buf[0]).astype(np.float64) / 100e3         # ms
dur = buf[1::2].astype(np.float64) / 100.0                    # us
print("probe: %d samples over %.0f ms, chain %.1f us at its fastest" % (samples, t[-1], dur.min()))
edges = np.arange(0, t[-1] + 4, 4.0)
idx = np.digitize(t, edges)
line = []
for b in range(1, len(edges)):
    m = idx == b
    if m.any():
        line.append("%.2f" % np.median(dur[m]))
print("median chain duration (us) per 4 ms:", " ".join(line))
