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
    na = len(host["q_arena"])
    code = np.full(256, 4, np.int64)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    prof = torch.randint(1, 30, (na, 11), dtype=torch.int16, device="cuda:0", generator=g)
    prof[torch.arange(na, device="cuda:0"), torch.from_numpy(code[host["q_arena"]]).to("cuda:0")] = -5
    dq = dict(d, q_profile=prof.reshape(-1))
    L = dict(align_type=_lib.ALIGN_LOCAL)
    for name, b, kw, wave_pairs in (("Global", d, {}, None), ("Local, lane classes", d, L, None), ("Local, one wave per pair", d, L, "4000000000"),
                                    ("Local with a quality-value profile, lane classes", dq, L, None),
                                    ("Local with a quality-value profile, one wave per pair", dq, L, "4000000000")):
        if wave_pairs:
            os.environ["GBA_WAVE_PAIRS"] = wave_pairs
        else:
            os.environ.pop("GBA_WAVE_PAIRS", None)
        ctx = api.Context(0, timing=True)
        ctx.set_scoring(api.make_scoring(**kw))
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
        print("%s: %d pairs, %.1f ms, %.1f GCUPS, ok %d, fill %.1f ms, one wave per pair %d" % (name, n, dt * 1e3, cells / dt / 1e9, int((r["status"] == 0).sum()), tm["fill_busy_ms"], int(tm["wide_pairs"])), flush=True)
        ctx.close()
    print("Local results equal on both kernel families:", digests[1] == digests[2], digests[3] == digests[4], flush=True)
    del d, dq, prof, out_pairs, out_blocks
    torch.cuda.empty_cache()
