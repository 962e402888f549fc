"""Device-resident batches through ONE context, and through TWO contexts on one GPU driven by two threads (the second
started half a batch later), same batch: does overlapping one batch's tail with the next one's head pay?"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("GBA_BENCH_HW_QUEUES", "16")
import torch
from blasr_amd import api, synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 491520
steps = 6
cfg = dict(synth.PRESETS["c2_10kb_linear"])
ref = synth.reference(cfg["ref_len"], synth.REF_SEED, 0.0)
host = synth.generate(n, ref=ref, **{a: b for a, b in cfg.items() if a != "ref"})
d = api.to_device(host)
cap = int(host["q_len"].astype(np.int64).sum() // 4) + n
ctxs, outs = [], []
for i in range(2):
    c = api.Context(0, timing=True)
    c.set_scoring(api.make_scoring(band=cfg["band"]))
    ctxs.append(c)
    outs.append((torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0"), torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0")))
for c, o in zip(ctxs, outs):
    c.refine_device(d, *o)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ctxs[0].refine_device(d, *outs[0])
torch.cuda.synchronize()
one = (time.perf_counter() - t0) / steps
cells = int(ctxs[0].timing()["cells"])
print("one context: %.1f ms per batch, %.1f GCUPS" % (one * 1e3, cells / one / 1e9), flush=True)
for stagger in (0.0, 0.5):
    def work(i):
        if i == 1 and stagger:
            time.sleep(stagger * one)
        for _ in range(steps):
            ctxs[i].refine_device(d, *outs[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / (2 * steps)
    same = outs[0][0].cpu().numpy().tobytes() == outs[1][0].cpu().numpy().tobytes()
    print("two contexts, second %.1f batches later: %.1f ms per batch, %.1f GCUPS, same results %s" % (stagger, two * 1e3, cells / two / 1e9, same), flush=True)
