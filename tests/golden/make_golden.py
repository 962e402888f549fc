#!/usr/bin/env python3
"""Generate tests/golden/*.json: small pairs with the oracle's outputs.

These fixtures pin ORACLE <-> KERNEL parity and guard the oracle against
regressions.  They do NOT pin parity with the BLASR binary: no output of the
reference exists in this environment (its alignment library is absent and all of
its test data lives on PacBio-internal mounts), so the expected values below come
from oracle/gba_oracle.c itself.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blasr_amd import synth          # noqa: E402
from oracle import oracle            # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

MICRO = [
    # name, q, t, blocks
    ("perfect", "ACGTACGTACGTAAGGCCTT", "ACGTACGTACGTAAGGCCTT", [(0, 0, 20)]),
    ("one_mismatch", "ACGTACGTACGAAAGGCCTT", "ACGTACGTACGTAAGGCCTT", [(0, 0, 11), (12, 12, 8)]),
    ("one_insertion", "ACGTACGTACGTTAAGGCCTT", "ACGTACGTACGTAAGGCCTT", [(0, 0, 11), (13, 12, 8)]),
    ("one_deletion", "ACGTACGTACGAAGGCCTT", "ACGTACGTACGTAAGGCCTT", [(0, 0, 11), (11, 12, 8)]),
    ("ins_next_to_del", "ACGTACGTAAACGTAGGCCTTAC", "ACGTACGTACGTCCAGGCCTTAC", [(0, 0, 8), (15, 15, 8)]),
    ("gap_wider_than_band", "ACGTACGTAC" + "GGCCTTAACC", "ACGTACGTAC" + "T" * 30 + "GGCCTTAACC", [(0, 0, 10), (10, 40, 10)]),
    ("n_bases", "ACGTNNGTACGTAAGGCCTT", "ACGTACGTNCGTAAGGCCTT", [(0, 0, 4), (9, 9, 11)]),
    ("lower_case", "acgtacgtacgtaaggcctt", "ACGTACGTACGTAAGGCCTT", [(0, 0, 20)]),
    ("tie_homopolymer", "AAAAAAAAAAAAA", "AAAAAAAAAAAAAAA", [(0, 0, 5), (8, 10, 5)]),
    ("tie_dinucleotide", "ACACACACACACAC", "ACACACACACAC", [(0, 0, 4), (10, 8, 4)]),
    ("offset_first_block", "TTACGTACGTACGTAAGG", "GGGACGTACGTACGTAAGG", [(2, 3, 8), (12, 13, 6)]),
    ("single_base_block", "A", "A", [(0, 0, 1)]),
]

PARAM_SETS = {
    "default": dict(),
    "tie_up_first": dict(tie_break=1),
    "band3": dict(band=3),
    "n_penalty": dict(n_score=2),
    "affine_stats": dict(stats_affine=1, stats_affine_open=4, stats_affine_extend=1),
}


def run(q, t, blocks, pkw):
    P = oracle.make_params(**pkw)
    res, blk = oracle.align(q, t, blocks, P)
    st, lo, hi, nc = oracle.guide_rows(blocks, len(q), len(t), P)
    return {"result": {k: int(v) for k, v in res.items()},
            "blocks": [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in blk],
            "guide_lo": [int(x) for x in lo], "guide_hi": [int(x) for x in hi]}


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = []
    for name, q, t, blocks in MICRO:
        for pname, pkw in PARAM_SETS.items():
            cases.append({"name": name + "/" + pname, "q": q, "t": t, "blocks": [list(b) for b in blocks],
                          "params": pkw, "expect": run(q, t, blocks, pkw)})
    json.dump({"note": "expected values produced by oracle/gba_oracle.c (parity with the BLASR binary unpinned)",
               "cases": cases}, open(os.path.join(OUT, "micro.json"), "w"), indent=0)
    # small synthetic pairs in the error model of the benchmark workload
    syn = []
    batch = synth.generate(24, read_len=260, len_jitter=120, seed=424242, threads=1)
    for i in range(24):
        q, t, blk = synth.pair(batch, i)
        q, t = bytes(q).decode(), bytes(t).decode()
        blocks = [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in blk]
        if not blocks:
            continue
        for pname in ("default", "tie_up_first"):
            pkw = PARAM_SETS[pname]
            e = run(q, t, blocks, pkw)
            del e["guide_lo"], e["guide_hi"]
            syn.append({"name": "synth%02d/%s" % (i, pname), "q": q, "t": t, "blocks": blocks,
                        "params": pkw, "expect": e})
    json.dump({"note": "expected values produced by oracle/gba_oracle.c (parity with the BLASR binary unpinned)",
               "cases": syn}, open(os.path.join(OUT, "synthetic_small.json"), "w"), indent=0)
    print(len(cases), "micro,", len(syn), "synthetic cases")


if __name__ == "__main__":
    main()
