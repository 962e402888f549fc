"""CPU tests of the oracle (oracle/gba_oracle.c): golden fixtures, hand-checkable micro cases
and the properties any correct banded global alignment must have."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases(name):
    return json.load(open(os.path.join(GOLD, name)))["cases"]


@pytest.mark.parametrize("fixture", ["micro.json", "synthetic_small.json"])
def test_oracle_matches_golden(oracle, fixture):
    for c in _cases(fixture):
        P = oracle.make_params(**c["params"])
        res, blk = oracle.align(c["q"], c["t"], c["blocks"], P)
        e = c["expect"]
        assert {k: int(v) for k, v in res.items()} == e["result"], c["name"]
        assert [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in blk] == e["blocks"], c["name"]
        if "guide_lo" in e:
            st, lo, hi, nc = oracle.guide_rows(c["blocks"], len(c["q"]), len(c["t"]), P)
            assert list(lo) == e["guide_lo"] and list(hi) == e["guide_hi"], c["name"]


def test_hand_checked_scores(oracle):
    P = oracle.make_params()          # match -5, mismatch 6, ins 5, del 4, rescoring indel 5
    r, b = oracle.align("ACGTACGTACGT", "ACGTACGTACGT", [(0, 0, 12)], P)
    assert (r["dp_score"], r["score"], r["n_match"]) == (-60, -60, 12) and len(b) == 1
    # one substitution: 11 matches + 1 mismatch
    r, b = oracle.align("ACGTACCTACGT", "ACGTACGTACGT", [(0, 0, 6), (7, 7, 5)], P)
    assert (r["dp_score"], r["n_match"], r["n_mismatch"]) == (-55 + 6, 11, 1)
    # one inserted query base: DP charges ins=5, rescoring charges indel=5
    r, b = oracle.align("ACGTACGGTACGT", "ACGTACGTACGT", [(0, 0, 6), (8, 7, 5)], P)
    assert (r["dp_score"], r["score"], r["n_ins"], r["n_del"]) == (-60 + 5, -60 + 5, 1, 0)
    # one deleted target base: DP charges del=4, rescoring charges indel=5
    r, b = oracle.align("ACGTACTACGT", "ACGTACGTACGT", [(0, 0, 6), (6, 7, 5)], P)
    assert (r["dp_score"], r["score"], r["n_ins"], r["n_del"]) == (-55 + 4, -55 + 5, 0, 1)


def test_statuses(oracle):
    P = oracle.make_params()
    assert oracle.align("ACGT", "ACGT", [], P)[0]["status"] == oracle.EMPTY_GUIDE
    assert oracle.align("ACGT", "ACGT", [(0, 0, 5)], P)[0]["status"] == oracle.BAD_GUIDE        # past the end
    assert oracle.align("ACGTACGT", "ACGTACGT", [(0, 0, 4), (3, 4, 2)], P)[0]["status"] == oracle.BAD_GUIDE  # overlap
    assert oracle.align("ACGTACGT", "ACGTACGT", [(0, 0, 0)], P)[0]["status"] == oracle.BAD_GUIDE


def test_tie_break_switch(oracle):
    # AAAAA vs AAAAAA...: several optimal paths; the two priorities must give the same optimum
    q, t, blk = "AAAAAAAAAAAAA", "AAAAAAAAAAAAAAA", [(0, 0, 5), (8, 10, 5)]
    r0, b0 = oracle.align(q, t, blk, oracle.make_params(tie_break=0))
    r1, b1 = oracle.align(q, t, blk, oracle.make_params(tie_break=1))
    assert r0["dp_score"] == r1["dp_score"]
    assert r0["n_match"] == r1["n_match"] == 13


def _path_cells(blocks, r):
    """(row, col) DP cells visited by the diagonal moves of a block list."""
    for b in blocks:
        for x in range(int(b["length"])):
            yield int(b["q_pos"]) + x + 1, int(b["t_pos"]) + x + 1


def test_properties_random(oracle, synth):
    batch = synth.generate(40, read_len=900, len_jitter=500, seed=99, threads=2)
    Plin = oracle.make_params()
    Psym = oracle.make_params(ins=5, dele=5, indel=5)
    for i in range(40):
        q, t, blk = synth.pair(batch, i)
        r, out = oracle.align(q, t, blk, Plin)
        assert r["status"] == 0
        # the true alignment stays within the band of this workload: banded == unbanded optimum
        assert r["dp_score"] == oracle.full_dp_score(q, t, Plin)
        # blocks: gap-free, strictly increasing, inside the region, and they cover the counts
        assert int(out["length"].sum()) == r["n_match"] + r["n_mismatch"]
        qe = out["q_pos"].astype(np.int64) + out["length"]
        te = out["t_pos"].astype(np.int64) + out["length"]
        assert np.all(out["q_pos"][1:] >= qe[:-1]) and np.all(out["t_pos"][1:] >= te[:-1])
        assert np.all((out["q_pos"][1:] > qe[:-1]) | (out["t_pos"][1:] > te[:-1]))   # merged runs
        assert qe[-1] <= r["rows"] and te[-1] <= r["cols"]
        assert r["n_match"] + r["n_mismatch"] + r["n_ins"] == r["rows"]
        assert r["n_match"] + r["n_mismatch"] + r["n_del"] == r["cols"]
        # path stays inside the band
        st, lo, hi, nc = oracle.guide_rows(blk, len(q), len(t), Plin)
        assert nc == r["ncells"]
        for row, col in _path_cells(out, r):
            assert lo[row] <= col <= hi[row]
        # with ins == del == indel the rescored path score is the DP optimum
        r2, _ = oracle.align(q, t, blk, Psym)
        assert r2["score"] == r2["dp_score"]


def test_guide_geometry(oracle, synth):
    batch = synth.generate(10, read_len=2000, seed=5, threads=1)
    P = oracle.make_params(band=10)
    for i in range(10):
        q, t, blk = synth.pair(batch, i)
        st, lo, hi, nc = oracle.guide_rows(blk, len(q), len(t), P)
        assert st == 0 and len(lo) == len(q) + 1
        assert lo[0] == 0 and hi[0] == 10
        assert np.all(np.diff(lo) >= 0), "first column never moves left"
        assert np.all(lo[1:] <= hi[:-1] + 1), "consecutive rows stay connected"
        assert np.all(hi <= len(t)) and hi[-1] == len(t)
        # every guide block lies inside the band
        for b in blk:
            for x in (0, int(b["length"]) - 1):
                row, col = int(b["q_pos"]) + x + 1, int(b["t_pos"]) + x + 1
                assert lo[row] <= col <= hi[row]


def test_kband_equals_full_dp(oracle, synth):
    batch = synth.generate(6, read_len=400, seed=11, threads=1)
    for i in range(6):
        q, t, _ = synth.pair(batch, i)
        k = abs(len(q) - len(t)) + 40
        P = oracle.make_params(mode=oracle.MODE_KBAND, kband_k=k)
        r, out = oracle.align(q, t, [], P)
        assert r["status"] == 0 and r["dp_score"] == oracle.full_dp_score(q, t, P)
        P0 = oracle.make_params(mode=oracle.MODE_KBAND, kband_k=max(abs(len(q) - len(t)) - 1, 0))
        if abs(len(q) - len(t)) > 0:
            assert oracle.align(q, t, [], P0)[0]["status"] == oracle.END_OUT_OF_BAND


def test_affine_model(oracle):
    # with gap costs 100 the linear moves are unusable and gaps must go through the affine states
    # (the reference's own affine test forces this the same way, ctest/affineAlign.t:6)
    P = oracle.make_params(mode=oracle.MODE_AFFINE, ins=100, dele=100, affine_open=10, affine_extend=1, band=16)
    q = "ACGTACGTACGT" + "GGG" + "TTGACCATGCAA"
    t = "ACGTACGTACGT" + "TTGACCATGCAA"
    r, b = oracle.align(q, t, [(0, 0, 12), (15, 12, 12)], P)
    assert r["status"] == 0 and r["n_ins"] == 3 and r["n_del"] == 0
    assert r["dp_score"] == -5 * 24 + 10 + 3 * 1
    # one long gap is cheaper than two short ones under open=10
    P2 = oracle.make_params(mode=oracle.MODE_AFFINE, ins=100, dele=100, affine_open=10, affine_extend=0, band=16)
    r2, _ = oracle.align(q, t, [(0, 0, 12), (15, 12, 12)], P2)
    assert r2["dp_score"] == -120 + 10


def test_quality_value_costs_in_both_gap_models(oracle, synth):
    """gbo_align_qv: a profile that spells out the matrix row of every base, ins and del gives the plain results, linear and
    affine; a hand-checked case where one base's quality makes its deletion cheap; profile values are taken within
    +-127 (linear) and +-63 (affine)."""
    batch = synth.generate(6, read_len=300, seed=5, threads=1)
    for mode, extra in ((oracle.MODE_LINEAR, {}), (oracle.MODE_AFFINE, dict(affine_open=6, affine_extend=2))):
        kw = dict(mode=mode, ins=5, dele=4, band=12, **extra)
        P = oracle.make_params(**kw)
        for i in range(6):
            q, t, blk = synth.pair(batch, i)
            code = {65: 0, 67: 1, 71: 2, 84: 3}
            prof = np.zeros((len(q), 11), np.int16)
            for x, c in enumerate(q):
                prof[x, :5] = [P.sub[code.get(int(c) & 0xdf, 4)][b] for b in range(5)]
            prof[:, 5] = 5
            prof[:, 6:] = 4
            a, ab = oracle.align(q, t, blk, P)
            b, bb = oracle.align_qv(q, t, blk, P, prof.reshape(-1))
            assert a == b and (ab == bb).all()
        # q = ACGT ACGT, t = ACGT G ACGT: the extra G is deleted behind query base 3; its cost comes from that base's entry
        q, t = b"ACGTACGT", b"ACGTGACGT"
        prof = np.zeros((8, 11), np.int16)
        prof[:, :5] = 6
        for x, c in enumerate(q):
            prof[x, b"ACGT".index(c)] = -5
        prof[:, 5] = 9
        prof[:, 6:] = 9
        prof[3, 6 + 2] = 1                                  # deleting a G behind query base 3
        r, blocks = oracle.align_qv(q, t, [(0, 0, 4), (4, 5, 4)], oracle.make_params(**dict(kw, affine_open=20)), prof.reshape(-1))
        assert r["status"] == 0 and r["dp_score"] == -40 + 1 and r["n_del"] == 1
        prof[:, 5:] = 500                                   # every gap move dearer than the limit: exactly one deletion is needed
        r, _ = oracle.align_qv(q, t, [(0, 0, 4), (4, 5, 4)], oracle.make_params(**dict(kw, affine_open=200, dele=120)), prof.reshape(-1))   # row 0 deletes at dele
        assert r["dp_score"] == -40 + (63 if mode == oracle.MODE_AFFINE else 127)
