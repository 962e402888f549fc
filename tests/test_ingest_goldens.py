"""The `blasr -m 5` ingest hook (tests/ingest_goldens.py) on records written from the oracle's own
alignments: parsing, block extraction and the comparison logic.  No real BLASR output exists here,
so this covers the hook, not parity with the binary."""
import numpy as np

import ingest_goldens as ig


def test_blocks_from_strings_round_trip():
    q, t = b"ACGTTTGACCA", b"ACGGACGTCA"
    blocks = [(0, 0, 3), (6, 3, 3), (9, 8, 2)]
    qa, ta = ig.strings_from_blocks(q, t, blocks)
    assert len(qa) == len(ta) and qa.replace("-", "").encode() == q and ta.replace("-", "").encode() == t
    q2, t2, b2 = ig.blocks_from_strings(qa, ta)
    assert (q2, t2, b2) == (q, t, blocks)
    assert ig.blocks_from_strings("AC--GT", "ACTT-T") == (b"ACGT", b"ACTTT", [(0, 0, 2), (3, 4, 1)])


def test_parse_m5_rejects_headers_and_ragged_lines():
    assert ig.parse_m5("") is None
    assert ig.parse_m5("qName qLength qStart qEnd qStrand tName tLength tStart tEnd tStrand score numMatch "
                       "numMismatch numIns numDel mapQV qAlignedSeq matchPattern tAlignedSeq") is None
    rec = ig.parse_m5("r/0 4 0 4 + ref 5 0 5 + -14 4 0 0 1 254 AC-GT ||*|| ACTGT")
    assert rec["score"] == -14 and rec["numDel"] == 1 and rec["tAlignedSeq"] == "ACTGT"


def test_oracle_alignments_survive_the_m5_round_trip(oracle, synth):
    batch = synth.generate(24, read_len=600, len_jitter=100, seed=77, min_block=11, drop_frac=0.1)
    P = oracle.make_params()
    res, blks = oracle.align_batch(batch, P)
    lines = []
    for p in range(int(batch["n_pairs"])):
        r = res[p]
        assert int(r["status"]) == 0
        qo, ql = int(batch["q_off"][p]), int(batch["q_len"][p])
        to, tl = int(batch["t_off"][p]), int(batch["t_len"][p])
        q = bytes(batch["q_arena"][qo:qo + ql]); t = bytes(batch["t_arena"][to:to + tl])
        blocks = [(int(b["q_pos"]) + int(r["q_pos"]), int(b["t_pos"]) + int(r["t_pos"]), int(b["length"]))
                  for b in blks[p]]
        lines.append(ig.format_m5("read/%d" % p, q, t, blocks, int(r["score"]),
                                  (int(r["n_match"]), int(r["n_mismatch"]), int(r["n_ins"]), int(r["n_del"]))))
    records = [ig.parse_m5(line) for line in lines]
    assert all(records)
    summary, rows = ig.check_records(records, oracle)
    assert summary["aligned"] == len(records) and summary["skipped"] == 0
    # the printed path lies inside the band around itself: re-aligning can only keep or improve the DP score
    assert summary["dp_worse"] == 0
    assert all(r["oracle_dp"] <= r["m5_path_dp"] for r in rows)
    # where the path is reproduced, so are the counts and the score
    for r in rows:
        if r["path_equal"]:
            assert r["counts_equal"] and r["score_equal"]
    assert summary["path_equal"] >= len(records) * 3 // 4

    # a corrupted record (one match column turned into a mismatch in the printed counts) is caught
    bad = dict(records[0]); bad["numMatch"] -= 1; bad["numMismatch"] += 1
    s2, _ = ig.check_records([bad], oracle)
    assert s2["counts_equal"] == 0
