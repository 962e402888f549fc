import numpy as np


def test_deterministic_and_valid(synth):
    a = synth.generate(16, read_len=1200, seed=3, threads=1)
    b = synth.generate(16, read_len=1200, seed=3, threads=4)
    for k in ("q_arena", "t_arena", "q_len", "t_len", "n_blk"):
        assert np.array_equal(a[k], b[k]), k
    for i in range(16):
        q, t, blk = synth.pair(a, i)
        assert len(blk) > 0 and blk[0]["q_pos"] == 0 and blk[0]["t_pos"] == 0
        for x in blk:                     # guide blocks are exact matches of at least the SDP word size
            assert x["length"] >= 11
            qs = q[x["q_pos"]:x["q_pos"] + x["length"]]
            ts = t[x["t_pos"]:x["t_pos"] + x["length"]]
            assert np.array_equal(qs, ts)
        last = blk[-1]
        assert last["q_pos"] + last["length"] == len(q) and last["t_pos"] + last["length"] == len(t)


def test_error_rates(synth, oracle):
    b = synth.generate(8, read_len=5000, seed=8, threads=2)
    P = oracle.make_params()
    res, _ = oracle.align_batch(b, P)
    nm, nx, ni, nd = (res[k].sum() for k in ("n_match", "n_mismatch", "n_ins", "n_del"))
    cols = nm + nx + ni + nd
    assert 0.10 < (nx + ni + nd) / cols < 0.20        # ~15 % total error
    assert ni > nd > nx                               # indel-heavy, insertions dominate


def test_reference_has_n_runs(synth):
    ref = synth.reference(200_000, seed=1, n_frac=0.05, n_run=500)
    frac = (ref == ord("N")).mean()
    assert 0.02 < frac < 0.06
