#!/usr/bin/env python3
"""Generate tests/golden/kband_small.json: small pairs with the outputs of oracle/kband_oracle.c for every
alignment type / gap model of the fixed-band aligners.  Like the other fixtures these pin the ORACLE
(against regressions) and kernel <-> oracle parity, not the BLASR binary: no output of the reference's
KBandAlign / AffineKBandAlign / ExtendAlignment exists in this environment."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from blasr_amd import api, _lib      # noqa: E402  (parameter structures only: no device is touched)
from kband_cases import make_pairs, random_profile   # noqa: E402
from oracle import oracle            # noqa: E402

SETS = [
    ("global", dict(align_type=_lib.KB_GLOBAL, ins=5, dele=4)),
    ("global_tie_up_first", dict(align_type=_lib.KB_GLOBAL, ins=5, dele=4, tie_break=1)),
    ("fit", dict(align_type=_lib.KB_FIT, ins=5, dele=4)),
    ("extend", dict(align_type=_lib.KB_EXTEND, ins=4, dele=5, max_drops=10)),
    ("extend_no_drops", dict(align_type=_lib.KB_EXTEND, ins=4, dele=5, max_drops=0)),
    ("affine_hp", dict(align_type=_lib.KB_GLOBAL, gap_model=_lib.KB_AFFINE_HP)),
    ("affine_hp_rule1", dict(align_type=_lib.KB_GLOBAL, gap_model=_lib.KB_AFFINE_HP, hp_rule=1)),
    ("global_profile", dict(align_type=_lib.KB_GLOBAL, ins=7, dele=7)),
]


def main():
    rng = np.random.default_rng(20250218)
    cases = []
    for name, kw in SETS:
        pairs, ks = make_pairs(rng, 6, (20, 160), flank=kw["align_type"] == _lib.KB_FIT, homopolymers="affine" in name)
        prof = random_profile(rng, pairs) if name.endswith("profile") else None
        P = api.make_kband_params(**kw)
        off = 0
        for x, ((q, t), k) in enumerate(zip(pairs, ks)):
            pr = None
            if prof is not None:
                pr = prof[off * 11:(off + len(q)) * 11]; off += len(q)
            rev = x % 3 == 2
            if kw["align_type"] == _lib.KB_EXTEND:
                k = 10
            res, blk = oracle.kband_align(q, t, k, P, profile=pr, reverse=rev)
            cases.append({"set": name, "params": kw, "q": bytes(q).decode(), "t": bytes(t).decode(), "k": int(k), "reverse": bool(rev),
                          "profile": [int(v) for v in pr] if pr is not None else None,
                          "result": {a: int(b) for a, b in res.items()},
                          "blocks": [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in blk]})
    with open(os.path.join(ROOT, "tests", "golden", "kband_small.json"), "w") as f:
        json.dump({"made_by": "tests/golden/make_kband_golden.py (oracle/kband_oracle.c; parity with BLASR unpinned)", "cases": cases}, f)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
