"""The guided DP under the Local end rule (gba_scoring.align_type GBA_ALIGN_LOCAL) on the lane classes against one wave per
pair (GBA_WAVE_PAIRS raised) and the Global rule, device-resident pairs of 10 kb.  usage: local_time.py pairs ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blasr_amd import api, synth, _lib

for n in [int(a) for a in sys.argv[1:]] or [122880]:
    host = synth.compact_blocks(synth.preset("c2_10kb_linear", n)[0])
    d = api.to_device(host)
    out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
    out_blocks = torch.zeros((int(host["q_len"].astype(np.int64).sum() // 4) + n, 3), dtype=torch.int32, device="cuda:0")
    digests = []
    for name, kw, wave_pairs in (("Global", {}, None), ("Local, lane classes", dict(align_type=_lib.ALIGN_LOCAL), None),
                                 ("Local, one wave per pair", dict(align_type=_lib.ALIGN_LOCAL), "4000000000")):
        if wave_pairs:
            os.environ["GBA_WAVE_PAIRS"] = wave_pairs
        else:
            os.environ.pop("GBA_WAVE_PAIRS", None)
        ctx = api.Context(0, timing=True)
        ctx.set_scoring(api.make_scoring(**kw))
        ctx.refine_device(d, out_pairs, out_blocks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            rc, needed = ctx.refine_device(d, out_pairs, out_blocks)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        r = out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT)
        cells = int(r["ncells"][r["status"] == 0].sum())
        tm = ctx.timing()
        digests.append(r.tobytes())
        print("%s: %d pairs, %.1f ms, %.1f GCUPS, ok %d, fill %.1f ms, one wave per pair %d" % (name, n, dt * 1e3, cells / dt / 1e9, int((r["status"] == 0).sum()), tm["fill_busy_ms"], int(tm["wide_pairs"])), flush=True)
        ctx.close()
    print("Local results equal on both kernel families:", digests[1] == digests[2], flush=True)
    del d, out_pairs, out_blocks
    torch.cuda.empty_cache()
