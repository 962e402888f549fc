"""The guided DP with and without a quality-value profile (gba_batch.q_profile), linear and affine gaps, device-resident, 32 768 pairs of 10 kb."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blasr_amd import api, synth, _lib

n = 32768
host = synth.compact_blocks(synth.preset("c2_10kb_linear", n)[0])
rng = np.random.default_rng(1)
na = len(host["q_arena"])
prof = rng.integers(1, 30, (na, 11), dtype=np.int16)
code = np.full(256, 4, np.int64)
for i, c in enumerate(b"ACGT"):
    code[c] = i
prof[np.arange(na), code[host["q_arena"]]] = -5
ctx = api.Context(0, timing=True)
ctx.set_scoring(api.make_scoring())
out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
out_blocks = torch.zeros((int(host["q_len"].astype(np.int64).sum() // 4) + n, 3), dtype=torch.int32, device="cuda:0")
aff = dict(mode=_lib.MODE_AFFINE, band=16, affine_open=6, affine_extend=2, stats_affine=1)
for name, b, kw in (("matrix scorer (wave-per-pair kernels forced: GBA_WAVE_PAIRS)", host, {}), ("quality-value profile", dict(host, q_profile=prof.reshape(-1)), {}),
                    ("affine gaps, matrix scorer (wave-per-pair kernels forced)", host, aff), ("affine gaps, quality-value profile", dict(host, q_profile=prof.reshape(-1)), aff)):
    ctx.set_scoring(api.make_scoring(**kw))
    d = api.to_device(b)
    ctx.refine_device(d, out_pairs, out_blocks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        rc, needed = ctx.refine_device(d, out_pairs, out_blocks)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    r = out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT)
    cells = int(r["ncells"][r["status"] == 0].sum())
    print("%s: %d pairs, %.1f ms, %.1f GCUPS, ok %d, fill %.1f ms" % (name, n, dt * 1e3, cells / dt / 1e9, int((r["status"] == 0).sum()), ctx.timing()["fill_ms"]), flush=True)
    del d
