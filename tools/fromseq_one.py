"""One context, one batch at a time: the from-sequences host path without a second context beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GBA_NO_TORCH", "1")
import numpy as np
from blasr_amd import api, synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 491520
cfg = dict(synth.PRESETS["c2_10kb_linear"])
ref = synth.reference(cfg["ref_len"], synth.REF_SEED, cfg.get("n_frac", 0.0))
host = synth.as_windows(synth.compact_blocks(synth.generate(n, ref=ref, **{a: b for a, b in cfg.items() if a != "ref"})))
ctx = api.Context(0)
ctx.set_scoring(api.make_scoring(band=cfg["band"]))
ctx.set_reference(ref)
pb = ctx.pinned_copy(host)
pairs = ctx.host_array(n, _lib.PAIR_RESULT_DT)
blocks = ctx.host_array(560_000_000 * n // 491520, _lib.BLOCK_DT)
seqs = {k: v for k, v in pb.items() if k not in ("blocks", "blk_off", "n_blk")}
ctx.set_guide_producer(api.make_sdp_params())
for name, b, out in (("guided, blocks out", pb, blocks), ("guided, intervals", pb, None), ("sequences, blocks out", seqs, blocks), ("sequences, intervals", seqs, None)):
    ts = []
    for i in range(4):
        t0 = time.perf_counter()
        ctx.submit_host(b, pairs, out); rc, needed = ctx.wait_host()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-24s rc %d ms %s" % (name, rc, " ".join("%.1f" % t for t in ts)), flush=True)
ctx.close()
