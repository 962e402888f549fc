"""The other BASELINE configurations streamed through api.DevicePipeline at depths 1..4: usage pipe_depth.py [batches]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("GBA_BENCH_HW_QUEUES", "16")
import torch
from blasr_amd import api, synth, _lib
import bench

count = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for name, n, band in (("c3_20kb_affine", 16384, None), ("c5_50kb_wide", 2048, 100), ("c1_ecoli_1kb", 4096, None)):
    if name not in synth.PRESETS:
        continue
    cfg = dict(synth.PRESETS[name])
    if band is not None:
        cfg["band"] = band
    k = cfg.get("decoy_every", 0) or 1
    n -= n % k
    ref = synth.reference(cfg["ref_len"], synth.REF_SEED, cfg.get("n_frac", 0.0))
    host = synth.generate(n, share_reads=k > 1, ref=ref, **{a: b for a, b in cfg.items() if a != "ref"})
    d = api.to_device(host)
    cap = int(host["q_len"].astype(np.int64).sum() // 4) + n
    for depth in (1, 2, 3, 4):
        pipe = api.DevicePipeline(0, depth=depth)
        pipe.set_scoring(api.make_scoring(band=cfg["band"], mode=_lib.MODE_AFFINE if cfg["mode"] == "affine" else _lib.MODE_LINEAR))
        bufs = [(torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0"), torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0")) for _ in range(depth)]
        dt, rc = bench.stream_batches(torch, pipe, d, bufs, depth, count)
        tmg = api.Context(0, timing=True)
        print("%s depth %d: %.2f ms per batch rc %d" % (name, depth, dt * 1e3, rc), flush=True)
        tmg.close()
        pipe.close()
        del bufs
        torch.cuda.empty_cache()
