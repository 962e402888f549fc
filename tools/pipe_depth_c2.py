"""The default workload streamed through api.DevicePipeline: usage pipe_depth_c2.py depth [batches]; GBA_SLOTS_DEVICE from the environment"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("GBA_BENCH_HW_QUEUES", "16")
import torch
from blasr_amd import api, synth, _lib
import bench

depth = int(sys.argv[1])
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = 491520
cfg = dict(synth.PRESETS["c2_10kb_linear"])
ref = synth.reference(cfg["ref_len"], synth.REF_SEED, 0.0)
host = synth.generate(n, ref=ref, **{a: b for a, b in cfg.items() if a != "ref"})
d = api.to_device(host)
cap = int(host["q_len"].astype(np.int64).sum() // 4) + n
del host
pipe = api.DevicePipeline(0, depth=depth)
pipe.set_scoring(api.make_scoring(band=cfg["band"]))
bufs = [(torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0"), torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0")) for _ in range(depth)]
dt, rc = bench.stream_batches(torch, pipe, d, bufs, depth, count)
free, total = torch.cuda.mem_get_info()
print("slots %s depth %d: %.2f ms per batch rc %d, device memory in use %.1f GB" % (os.environ.get("GBA_SLOTS_DEVICE", "3"), depth, dt * 1e3, rc, (total - free) / 1e9), flush=True)
pipe.close()
