"""Is the length of a one-part batch's traceback tail a property of the context?  Several contexts alive at once, each
with its own scratch and streams, run the same device-resident batch (122 880 pairs of 10 kb, Local end rule) in turn,
three rounds: per context and round the step, the fill and the tail (gba_timing).  usage: tail_modes.py [contexts = 5] [pairs = 122880]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # sets GPU_MAX_HW_QUEUES before the runtime starts
import numpy as np
import torch
from blasr_amd import api, synth, _lib

k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 122880
cfg = dict(synth.PRESETS["c2_10kb_linear"])
ref = synth.reference(cfg["ref_len"], synth.REF_SEED, cfg.get("n_frac", 0.0))
host = synth.generate(n, ref=ref, **{a: b for a, b in cfg.items() if a != "ref"})
d = api.to_device(host, "cuda:0")
cap = int(host["q_len"].astype(np.int64).sum() // 4) + n
out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
out_blocks = torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0")
ctxs = []
for i in range(k):
    c = api.Context(0, timing=True)
    c.set_scoring(api.make_scoring(band=cfg["band"], align_type=_lib.ALIGN_LOCAL))
    c.refine_device(d, out_pairs, out_blocks)
    ctxs.append(c)
torch.cuda.synchronize()
for rnd in range(3):
    line = []
    for c in ctxs:
        t0 = time.perf_counter()
        c.refine_device(d, out_pairs, out_blocks)
        torch.cuda.synchronize()
        tm = c.timing()
        line.append("%.1f/%.1f/%.1f" % ((time.perf_counter() - t0) * 1e3, tm["fill_busy_ms"], tm["traceback_ms"]))
    print("round", rnd, " ".join(line), flush=True)
for c in ctxs:
    c.close()
