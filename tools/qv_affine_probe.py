"""Which pairs of a quality-value batch with affine gaps differ from the oracle, by band and profile sign."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from blasr_amd import api, synth, _lib
from oracle import oracle

ctx = api.Context(0)
for band in (40, 70, 150):
    for neg in (False, True):
        rng = np.random.default_rng(band)
        batch = synth.compact_blocks(synth.generate(48, read_len=800, len_jitter=500, seed=3 + band))
        n = len(batch["q_arena"])
        prof = rng.integers(0, 25, (n, 11)).astype(np.int16)
        prof[:, :5] += 5
        if neg:
            prof[rng.random(n) < 0.05, 7] = -60
        prof = prof.reshape(-1)
        kw = dict(mode=_lib.MODE_AFFINE, band=band, ins=5, dele=4, affine_open=3, affine_extend=2, stats_affine=1)
        ctx.set_scoring(api.make_scoring(**kw))
        pairs, blocks, rc = ctx.refine_host(dict(batch, q_profile=prof))
        P = oracle.make_params(**kw)
        bad = []
        for p in range(48):
            q, t, blk = synth.pair(batch, p)
            o = int(batch["q_off"][p])
            want, wb = oracle.align_qv(q, t, blk, P, prof[o * 11:(o + len(q)) * 11])
            if int(pairs[p]["dp_score"]) != want["dp_score"]:
                bad.append((p, int(pairs[p]["dp_score"]) - want["dp_score"]))
        print("band", band, "negatives", neg, "bad", len(bad), bad[:6], flush=True)
