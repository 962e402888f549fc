"""The CPU restatement of the fixed-band aligners (oracle/kband_oracle.c) against properties any
correct implementation has and against hand-checkable cases.  No GPU."""
import numpy as np
import pytest

from kband_cases import make_pairs, random_profile


@pytest.fixture(scope="module")
def P():
    from blasr_amd import api, _lib
    return api, _lib


def _u8(s):
    return np.frombuffer(s.encode(), np.uint8).copy()


def test_hand_cases(oracle, P):
    api, L = P
    g = api.make_kband_params(align_type=L.KB_GLOBAL, ins=5, dele=4)
    r, b = oracle.kband_align(_u8("ACGTACGTAC"), _u8("ACGTACGTAC"), 3, g)
    assert (r["status"], r["dp_score"], r["n_match"], r["n_blk"]) == (0, -50, 10, 1) and tuple(b[0]) == (0, 0, 10)
    r, b = oracle.kband_align(_u8("ACGTTACGTAC"), _u8("ACGTACGTAC"), 3, g)       # one inserted base
    assert (r["dp_score"], r["n_ins"], r["n_del"], r["n_blk"]) == (-45, 1, 0, 2)
    r, b = oracle.kband_align(_u8("ACGTACGTAC"), _u8("ACGTTACGTAC"), 3, g)       # one deleted base
    assert (r["dp_score"], r["n_ins"], r["n_del"]) == (-46, 0, 1)
    r, _ = oracle.kband_align(_u8("ACGT"), _u8("ACGTACGTAC"), 3, g)
    assert r["status"] == oracle.END_OUT_OF_BAND
    f = api.make_kband_params(align_type=L.KB_FIT, ins=5, dele=4)
    r, b = oracle.kband_align(_u8("GGGTTT"), _u8("ACAGGGTTTCA"), 6, f)           # the query sits at target 3..9
    assert (r["dp_score"], r["t_pos"], r["q_pos"], r["n_blk"]) == (-30, 3, 0, 1) and tuple(b[0]) == (0, 0, 6)


def test_band_that_covers_everything_equals_the_full_dp(oracle, P):
    api, L = P
    rng = np.random.default_rng(7)
    for at in (L.KB_GLOBAL, L.KB_FIT):
        pairs, _ = make_pairs(rng, 12, (20, 200), flank=at == L.KB_FIT)
        prm = api.make_kband_params(align_type=at, ins=5, dele=4)
        for q, t in pairs:
            r, b = oracle.kband_align(q, t, len(q) + len(t), prm)
            assert r["status"] == 0 and r["dp_score"] == oracle.kband_full_dp(q, t, prm)
            # the path accounts for every query base, and for every target base when global
            assert r["n_match"] + r["n_mismatch"] + r["n_ins"] == len(q)
            if at == L.KB_GLOBAL:
                assert r["n_match"] + r["n_mismatch"] + r["n_del"] == len(t)
            assert int(b["length"].sum()) == r["n_match"] + r["n_mismatch"]


def test_rescoring_with_the_dp_costs_gives_the_dp_score(oracle, P):
    api, L = P
    rng = np.random.default_rng(9)
    pairs, ks = make_pairs(rng, 10, (50, 300))
    prm = api.make_kband_params(align_type=L.KB_GLOBAL, ins=5, dele=5, indel=5)
    for (q, t), k in zip(pairs, ks):
        r, _ = oracle.kband_align(q, t, k, prm)
        assert r["score"] == r["dp_score"]


def test_reverse_is_the_mirror_image(oracle, P):
    api, L = P
    rng = np.random.default_rng(13)
    pairs, ks = make_pairs(rng, 10, (50, 300))
    prof = random_profile(rng, pairs)
    off = 0
    for at in (L.KB_GLOBAL, L.KB_EXTEND):
        prm = api.make_kband_params(align_type=at, ins=5, dele=4, max_drops=4)
        off = 0
        for (q, t), k in zip(pairs, ks):
            pr = prof[off * 11:(off + len(q)) * 11]; off += len(q)
            a, ab = oracle.kband_align(q, t, k, prm, profile=pr)
            rp = pr.reshape(-1, 11)[::-1].reshape(-1).copy()
            m, mb = oracle.kband_align(q[::-1].copy(), t[::-1].copy(), k, prm, profile=rp, reverse=True)
            for f in ("status", "dp_score", "score", "n_blk", "n_match", "n_mismatch", "n_ins", "n_del", "ncells"):
                assert a[f] == m[f], f
            if len(ab):
                # block x of the mirrored run is block n-1-x of the forward run, seen from the other end
                qe = a["q_pos"] + int(ab[-1]["q_pos"]) + int(ab[-1]["length"])
                assert m["q_pos"] == len(q) - qe
                assert (mb["length"] == ab["length"][::-1]).all()


def test_extension_stops_where_the_sequences_part(oracle, P):
    api, L = P
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    common = rng.choice(acgt, 150)
    q = np.concatenate([common, rng.choice(acgt, 300)])
    t = np.concatenate([common, rng.choice(acgt, 300)])
    prm = api.make_kband_params(align_type=L.KB_EXTEND, ins=5, dele=4, max_drops=10)
    r, b = oracle.kband_align(q, t, 10, prm)
    assert r["status"] == 0 and r["dp_score"] <= -5 * 150 + 40
    end = r["q_pos"] + int(b[-1]["q_pos"]) + int(b[-1]["length"])
    assert 150 <= end <= 175
    assert r["ncells"] < 21 * 200                      # the fill stopped a few rows behind the best cell
    r0, b0 = oracle.kband_align(rng.choice(acgt, 50), np.full(50, ord("N"), np.uint8), 10,
                                api.make_kband_params(align_type=L.KB_EXTEND, n_score=3))
    assert (r0["status"], r0["dp_score"], r0["n_blk"]) == (0, 0, 0) and len(b0) == 0


def test_homopolymer_insertions_are_cheaper(oracle, P):
    api, L = P
    prm = api.make_kband_params(align_type=L.KB_GLOBAL, gap_model=L.KB_AFFINE_HP, indel=5)
    # three extra A behind an A: open 7, then 2 + 2 as a homopolymer run; 7 + 4 + 4 as a plain run behind a C
    hp, _ = oracle.kband_align(_u8("CCAAAAGG"), _u8("CCAGG"), 4, prm)
    pl, _ = oracle.kband_align(_u8("CCTTTCGG"), _u8("CCCGG"), 4, prm)
    assert hp["dp_score"] == -25 + 7 + 2 + 2 and hp["n_ins"] == 3
    assert pl["dp_score"] == -25 + 7 + 4 + 4 and pl["n_ins"] == 3


def _golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "kband_small.json")) as f:
        return json.load(f)["cases"]


def test_oracle_matches_golden(oracle, P):
    api, L = P
    cases = _golden()
    assert len(cases) >= 40
    for c in cases:
        prm = api.make_kband_params(**c["params"])
        pr = np.array(c["profile"], np.int16) if c["profile"] is not None else None
        r, b = oracle.kband_align(c["q"].encode(), c["t"].encode(), c["k"], prm, profile=pr, reverse=c["reverse"])
        assert {a: int(v) for a, v in r.items()} == c["result"], c["set"]
        assert [[int(x["q_pos"]), int(x["t_pos"]), int(x["length"])] for x in b] == c["blocks"], c["set"]
