"""The fixed-band aligners on the GPU (gba_kband, through the C-ABI) against their CPU restatement
oracle/kband_oracle.c, bit for bit: KBandAlign Global / Fit with the matrix scorer and with a
per-base cost profile (the quality-value scorers), AffineKBandAlign's insertion and
homopolymer-insertion states, and the banded extension with its drop-off, forwards and backwards.
Parity with libcpp's routines is unpinned (see the oracle's header)."""
import numpy as np
import pytest

from kband_cases import FIELDS, make_pairs, pack, random_profile

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from blasr_amd import api as a
    return a


def _run_and_compare(api, oracle, ctx, pairs, ks, P, profile=None, reverse=None):
    from blasr_amd import _lib
    batch = pack(pairs, ks, profile=profile, reverse=reverse)
    got, blocks, rc = ctx.kband_host(batch, P)
    assert rc == 0
    n = len(pairs)
    off = 0
    for p in range(n):
        q, t = pairs[p]
        prof = None
        if profile is not None:
            o = int(batch["q_off"][p])
            prof = batch["profile"][o * 11:(o + len(q)) * 11]
        want, wb = oracle.kband_align(q, t, ks[p], P, profile=prof, reverse=bool(reverse[p]) if reverse is not None else False)
        g = got[p]
        assert int(g["status"]) == want["status"], (p, int(g["status"]), want["status"])
        if want["status"] != 0:
            continue
        for f in FIELDS:
            assert int(g[f]) == int(want[f]), (f, p, int(g[f]), int(want[f]))
        assert int(g["blk_off"]) == off, (p, int(g["blk_off"]), off)
        kb = blocks[off:off + int(g["n_blk"])]
        off += int(g["n_blk"])
        assert (kb == wb).all(), p
        if len(wb):
            assert int(g["q_start"]) == want["q_pos"] and int(g["t_start"]) == want["t_pos"]
            assert int(g["q_end"]) == want["q_pos"] + int(wb[-1]["q_pos"]) + int(wb[-1]["length"])
            assert int(g["t_end"]) == want["t_pos"] + int(wb[-1]["t_pos"]) + int(wb[-1]["length"])
    assert off == len(blocks)
    return got


@pytest.mark.parametrize("tie", [0, 1])
@pytest.mark.parametrize("align_type", ["global", "fit"])
def test_kband_linear_matrix_scorer(api, oracle, align_type, tie):
    from blasr_amd import _lib
    rng = np.random.default_rng(11 + tie)
    ctx = api.Context(0)
    at = _lib.KB_GLOBAL if align_type == "global" else _lib.KB_FIT
    pairs, ks = make_pairs(rng, 48, (40, 900), flank=at == _lib.KB_FIT)
    # a pair whose end cell lies outside the band, an empty query, an empty target, a band of 0
    pairs += [(pairs[0][0], pairs[0][1][: len(pairs[0][1]) // 2]), (np.zeros(0, np.uint8), pairs[1][1][:5]),
              (pairs[2][0][:4], np.zeros(0, np.uint8)), (pairs[3][0][:50], pairs[3][0][:50])]
    ks += [3, 8, 8, 0]
    P = api.make_kband_params(align_type=at, ins=5, dele=4, tie_break=tie)
    got = _run_and_compare(api, oracle, ctx, pairs, ks, P)
    if at == _lib.KB_GLOBAL:
        assert int(got[48]["status"]) == _lib.PAIR_END_OUT_OF_BAND
    ctx.close()


def test_kband_wide_bands_use_several_diagonals_per_lane(api, oracle):
    from blasr_amd import _lib
    rng = np.random.default_rng(5)
    ctx = api.Context(0)
    for k in (40, 100, 250):
        pairs, _ = make_pairs(rng, 6, (300, 1200))
        P = api.make_kband_params(align_type=_lib.KB_GLOBAL, ins=5, dele=5)
        ks = [max(k, abs(len(q) - len(t))) for q, t in pairs]
        got = _run_and_compare(api, oracle, ctx, pairs, ks, P)
        # the band covers the optimal path: the banded optimum is the full-DP optimum
        for p, (q, t) in enumerate(pairs):
            if ks[p] >= 200:
                assert int(got[p]["dp_score"]) == oracle.kband_full_dp(q, t, P)
    too_wide, rc = ctx.kband_host(pack(pairs[:1], [256]), P)[0], 0
    assert int(too_wide[0]["status"]) == _lib.PAIR_TOO_WIDE
    ctx.close()


@pytest.mark.parametrize("align_type", ["global", "fit"])
def test_kband_quality_value_profile(api, oracle, align_type):
    from blasr_amd import _lib
    rng = np.random.default_rng(23)
    ctx = api.Context(0)
    at = _lib.KB_GLOBAL if align_type == "global" else _lib.KB_FIT
    pairs, ks = make_pairs(rng, 40, (30, 700), flank=at == _lib.KB_FIT)
    prof = random_profile(rng, pairs)
    P = api.make_kband_params(align_type=at, ins=7, dele=7)
    _run_and_compare(api, oracle, ctx, pairs, ks, P, profile=prof)
    ctx.close()


@pytest.mark.parametrize("hp_rule", [0, 1])
def test_affine_kband_homopolymer_insertions(api, oracle, hp_rule):
    from blasr_amd import _lib
    rng = np.random.default_rng(31 + hp_rule)
    ctx = api.Context(0)
    pairs, ks = make_pairs(rng, 40, (30, 800), homopolymers=True)
    for tie in (0, 1):
        P = api.make_kband_params(align_type=_lib.KB_GLOBAL, gap_model=_lib.KB_AFFINE_HP, hp_rule=hp_rule, tie_break=tie)
        got = _run_and_compare(api, oracle, ctx, pairs, ks, P)
    assert int(got["n_ins"].sum()) > 0
    ctx.close()


def test_extension_forwards_and_backwards(api, oracle):
    from blasr_amd import _lib
    rng = np.random.default_rng(41)
    ctx = api.Context(0)
    # sequences that agree for a stretch and then part ways: the extension must stop where they do
    pairs, ks, reverse = [], [], []
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for x in range(60):
        common = int(rng.integers(0, 400))
        base, _ = make_pairs(rng, 1, (max(common, 1), max(common, 1) + 1))
        q, t = base[0]
        q = np.concatenate([q[:common], rng.choice(acgt, int(rng.integers(0, 300)))])
        t = np.concatenate([t[:common], rng.choice(acgt, int(rng.integers(0, 300)))])
        rev = x % 2
        if rev:
            q, t = q[::-1].copy(), t[::-1].copy()
        pairs.append((q[:500], t[:500])); ks.append(10); reverse.append(rev)
    for drops in (10, 0, 3):
        P = api.make_kband_params(align_type=_lib.KB_EXTEND, ins=5, dele=4, max_drops=drops)
        got = _run_and_compare(api, oracle, ctx, pairs, ks, P, reverse=np.array(reverse, np.uint8))
    assert (got["dp_score"] <= 0).all()
    # other bands: up to 16 the extensions run one lane per pair (two register-window sizes), beyond that one wave per pair;
    # mixed bands in one batch; both tie-breaks
    P = api.make_kband_params(align_type=_lib.KB_EXTEND, ins=5, dele=4, max_drops=5, tie_break=1)
    for kk in (0, 1, 7, 11, 16, 17):
        _run_and_compare(api, oracle, ctx, pairs, [kk] * len(pairs), P, reverse=np.array(reverse, np.uint8))
    _run_and_compare(api, oracle, ctx, pairs, [int(x) for x in rng.integers(0, 17, len(pairs))], P, reverse=np.array(reverse, np.uint8))
    long_q = np.concatenate([pairs[0][0]] * 20)[:4200]
    r = ctx.kband_host(pack([(long_q, long_q)], [10]), P)[0]
    assert int(r[0]["status"]) == _lib.PAIR_TOO_LONG
    ctx.close()


def test_block_arena_too_small_and_no_arena(api, oracle):
    from blasr_amd import _lib
    rng = np.random.default_rng(3)
    ctx = api.Context(0)
    pairs, ks = make_pairs(rng, 8, (200, 400))
    P = api.make_kband_params()
    batch = pack(pairs, ks)
    full, blocks, rc = ctx.kband_host(batch, P)
    assert rc == 0
    need = int(full["n_blk"].sum())
    part, pblocks, rc = ctx.kband_host(batch, P, blocks_capacity=need // 2)
    assert rc == _lib.E_NOSPACE
    fits = np.cumsum(full["n_blk"]) <= need // 2
    assert ((part["status"] == _lib.PAIR_NOSPACE) == ~fits).all()
    assert (pblocks[: int(full["n_blk"][fits].sum())] == blocks[: int(full["n_blk"][fits].sum())]).all()
    ctx.close()


def test_golden_fixture_through_the_library(api, oracle):
    """tests/golden/kband_small.json (the oracle's outputs, committed) against the kernels."""
    import json
    import os
    from blasr_amd import _lib
    with open(os.path.join(os.path.dirname(__file__), "golden", "kband_small.json")) as f:
        cases = json.load(f)["cases"]
    ctx = api.Context(0)
    for c in cases:
        q, t = np.frombuffer(c["q"].encode(), np.uint8), np.frombuffer(c["t"].encode(), np.uint8)
        prof = np.array(c["profile"], np.int16) if c["profile"] is not None else None
        got, blocks, rc = ctx.kband_host(pack([(q, t)], [c["k"]], profile=prof, reverse=np.array([c["reverse"]], np.uint8)),
                                         api.make_kband_params(**c["params"]))
        assert rc == 0
        for f_ in FIELDS:
            assert int(got[0][f_]) == c["result"][f_], (c["set"], f_)
        assert [[int(x["q_pos"]), int(x["t_pos"]), int(x["length"])] for x in blocks] == c["blocks"], c["set"]
    ctx.close()


def test_full_length_pairs(api, oracle):
    """Reads of 10 kb in a band of 60 with a cost profile, and one pair of 50 kb in the widest band (k = 250)."""
    from blasr_amd import _lib
    rng = np.random.default_rng(77)
    ctx = api.Context(0)
    from kband_cases import ACGT, _read
    pairs = []
    for _ in range(6):              # 6 % errors: the lengths differ by about 100, inside the kernels' widest band of 255
        t = rng.choice(ACGT, int(rng.integers(9000, 10001))).astype(np.uint8)
        pairs.append((_read(rng, t, 0.06, False), t))
    ks = [max(60, abs(len(q) - len(t)) + 20) for q, t in pairs]
    assert max(ks) <= 255
    prof = random_profile(rng, pairs)
    _run_and_compare(api, oracle, ctx, pairs, ks, api.make_kband_params(align_type=_lib.KB_GLOBAL, ins=6, dele=6), profile=prof)
    tb = rng.choice(ACGT, 50000).astype(np.uint8)
    big = [(_read(rng, tb, 0.02, False), tb)]
    if abs(len(big[0][0]) - len(big[0][1])) <= 250:
        got = _run_and_compare(api, oracle, ctx, big, [250], api.make_kband_params(align_type=_lib.KB_GLOBAL, gap_model=_lib.KB_AFFINE_HP))
        assert int(got[0]["status"]) == 0 and int(got[0]["rows"]) == len(big[0][0])
    ctx.close()
