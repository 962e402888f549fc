"""The guided DP (linear gaps) with a quality-value profile on the lane-per-pair kernel (k_fill<false, true>) against the
wave-per-pair kernels (GBA_WAVE_PAIRS raised) and the matrix scorer, device-resident.  usage: qv_lane_time.py [affine] pairs ...
(affine: band 16, open 6, extend 2, k_fill<true, true>)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blasr_amd import api, synth, _lib

AFFINE = len(sys.argv) > 1 and sys.argv[1] == "affine"
KW = dict(mode=_lib.MODE_AFFINE, band=16, affine_open=6, affine_extend=2, stats_affine=1) if AFFINE else {}
ENV = "GBA_WAVE_PAIRS_AFFINE" if AFFINE else "GBA_WAVE_PAIRS"
for n in [int(a) for a in sys.argv[1:] if a != "affine"] or [32768]:
    host = synth.compact_blocks(synth.preset("c2_10kb_linear", n)[0])
    d = api.to_device(host)
    na = len(host["q_arena"])
    code = np.full(256, 4, np.int64)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    prof = torch.randint(1, 30, (na, 11), dtype=torch.int16, device="cuda:0", generator=g)
    prof[torch.arange(na, device="cuda:0"), torch.from_numpy(code[host["q_arena"]]).to("cuda:0")] = -5
    dp = dict(d, q_profile=prof.reshape(-1))
    out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
    out_blocks = torch.zeros((int(host["q_len"].astype(np.int64).sum() // 4) + n, 3), dtype=torch.int32, device="cuda:0")
    digests = []
    for name, b, wave_pairs in (("matrix scorer", d, None), ("quality-value profile, the library's choice", dp, None),
                                ("quality-value profile, lane classes whatever the batch size", dp, "1"),
                                ("quality-value profile, one wave per pair", dp, "4000000000")):
        if wave_pairs:
            os.environ[ENV] = wave_pairs
        else:
            os.environ.pop(ENV, None)
        ctx = api.Context(0, timing=True)
        ctx.set_scoring(api.make_scoring(**KW))
        ctx.refine_device(b, out_pairs, out_blocks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            rc, needed = ctx.refine_device(b, out_pairs, out_blocks)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        r = out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT)
        cells = int(r["ncells"][r["status"] == 0].sum())
        tm = ctx.timing()
        digests.append(r.tobytes())
        print("%s: %d pairs, %.1f ms, %.1f GCUPS, ok %d, fill %.1f ms, one wave per pair %d" % (name, n, dt * 1e3, cells / dt / 1e9, int((r["status"] == 0).sum()), tm["fill_ms"], int(tm["wide_pairs"])), flush=True)
        ctx.close()
    print("profile results equal on both kernel families:", digests[1] == digests[2] == digests[3], flush=True)
    del d, dp, prof, out_pairs, out_blocks
    torch.cuda.empty_cache()
