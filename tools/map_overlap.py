"""Sequences -> guides -> refined alignments with the two stages overlapped on one GPU: two contexts, two host
threads; while chunk i is refined, the guides of chunk i + 1 are made.  Prints the serial and the overlapped time."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from blasr_amd import api, synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 491520
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
batch, cfg = synth.preset("c2_10kb_linear", n)
per = n // chunks
from blasr_amd import shard
subs = [api.to_device(shard.sub_batch(batch, np.arange(i * per, (i + 1) * per))) for i in range(chunks)]
ctx_g, ctx_r = api.Context(0), api.Context(0)
ctx_r.set_scoring(api.make_scoring(band=cfg["band"]))
sp = api.make_sdp_params()
cap = int(batch["q_len"][:per].astype(np.int64).sum() // 4) + per
out_pairs = [torch.zeros(per * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
out_blocks = [torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0") for _ in range(2)]

def guides(i):
    g = ctx_g.sdp_device(subs[i], sp, blocks_capacity=cap)
    assert g["rc"] == 0
    return dict(subs[i], blocks=g["blocks"], blk_off=g["blk_off"], n_blk=g["n_blk"])

def refine(d, k):
    rc, _ = ctx_r.refine_device(d, out_pairs[k], out_blocks[k])
    assert rc == 0

def serial():
    for i in range(chunks):
        refine(guides(i), 0)

def overlapped():
    d = guides(0)
    for i in range(chunks):
        nxt = [None]
        th = None
        if i + 1 < chunks:
            th = threading.Thread(target=lambda j=i + 1: nxt.__setitem__(0, guides(j)))
            th.start()
        refine(d, i & 1)
        if th:
            th.join()
            d = nxt[0]

for name, fn in (("serial", serial), ("overlapped", overlapped), ("serial", serial), ("overlapped", overlapped)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-10s %d pairs in %d chunks: %.1f ms  %.2f M reads/s" % (name, n, chunks, dt * 1e3, n / dt / 1e6), flush=True)
