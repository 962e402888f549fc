"""Shared helpers of the parity tests: run a batch through the oracle and through the
C-ABI and compare every field RefineAlignment copies back."""
import numpy as np

FIELDS = ("status", "dp_score", "score", "q_pos", "t_pos", "n_blk", "n_match", "n_mismatch",
          "n_ins", "n_del", "rows", "cols", "ncells")


def oracle_params(oracle, **kw):
    return oracle.make_params(**kw)


def compare_batch(oracle, batch, oparams, pairs, blocks, expect_status=None):
    """Assert bit-equality of kernel results (pairs, blocks) with the oracle on `batch`."""
    ores, oblks = oracle.align_batch(batch, oparams)
    n = int(batch["n_pairs"])
    assert len(pairs) == n
    for p in range(n):
        o, k = ores[p], pairs[p]
        if expect_status is not None:
            assert int(k["status"]) == expect_status[p], (p, int(k["status"]), expect_status[p])
        assert int(o["status"]) == int(k["status"]), ("status", p, int(o["status"]), int(k["status"]))
        if int(o["status"]) != 0:
            continue
        for f in FIELDS:
            assert int(o[f]) == int(k[f]), (f, p, int(o[f]), int(k[f]))
        kb = blocks[int(k["blk_off"]):int(k["blk_off"]) + int(k["n_blk"])]
        ob = oblks[p]
        assert len(kb) == len(ob)
        assert np.array_equal(kb["q_pos"], ob["q_pos"]), ("q_pos", p)
        assert np.array_equal(kb["t_pos"], ob["t_pos"]), ("t_pos", p)
        assert np.array_equal(kb["length"], ob["length"]), ("length", p)
    return ores, oblks


def make_batch(pairs_list):
    """Build a batch dict from a list of (q bytes, t bytes, [(q_pos,t_pos,len),...])."""
    from blasr_amd.synth import BLOCK_DT
    n = len(pairs_list)
    q_off = np.zeros(n, np.uint64); t_off = np.zeros(n, np.uint64); b_off = np.zeros(n, np.uint64)
    q_len = np.zeros(n, np.uint32); t_len = np.zeros(n, np.uint32); n_blk = np.zeros(n, np.uint32)
    qa, ta, ba = bytearray(), bytearray(), []
    for i, (q, t, blk) in enumerate(pairs_list):
        q = bytes(q if not isinstance(q, str) else q.encode())
        t = bytes(t if not isinstance(t, str) else t.encode())
        q_off[i], t_off[i], b_off[i] = len(qa), len(ta), len(ba)
        q_len[i], t_len[i], n_blk[i] = len(q), len(t), len(blk)
        qa += q; ta += t; ba += [tuple(int(v) for v in b) for b in blk]
    blocks = np.zeros(max(len(ba), 1), dtype=BLOCK_DT)
    for i, b in enumerate(ba):
        blocks[i] = b
    return {"n_pairs": n,
            "q_arena": np.frombuffer(bytes(qa) + b"\0", dtype=np.uint8).copy(),
            "t_arena": np.frombuffer(bytes(ta) + b"\0", dtype=np.uint8).copy(),
            "blocks": blocks, "q_off": q_off, "t_off": t_off, "blk_off": b_off,
            "q_len": q_len, "t_len": t_len, "n_blk": n_blk}
