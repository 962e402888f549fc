"""The guided DP's Local end rule (gba_scoring.align_type = GBA_ALIGN_LOCAL: the GuidedAlign(..., Local, ...) call of
AlignSubreadToAlignmentTarget, /root/reference/iblasr/BlasrAlignImpl.hpp:1418-1421): free start, end in the cheapest cell
of the band.  CPU: the oracle's restatement against the unbanded local optimum and hand cases.  GPU: the wave-per-pair
kernels against the oracle, bit for bit, on reads whose ends do not align."""
import numpy as np
import pytest

import util

ACGT = np.frombuffer(b"ACGT", np.uint8)


def _noisy(rng, seq, rate):
    out = []
    for c in seq:
        r = rng.random()
        if r < rate * 0.5:
            out.append(int(rng.choice(ACGT))); out.append(c)
        elif r < rate * 0.8:
            continue
        elif r < rate:
            out.append(int(rng.choice(ACGT)))
        else:
            out.append(c)
    return bytes(out)


def _case(rng, core_len, junk_q, junk_t, rate=0.1):
    """A read and a window that share a noisy core; what lies in front of and behind it is unrelated sequence.  The guide
    (one exact block in each junk end, sparse exact blocks inside the core) spans everything, so the DP region holds ends
    that do not align."""
    import difflib
    core_t = rng.choice(ACGT, core_len).tobytes()
    core_q = _noisy(rng, core_t, rate)
    jq0, jq1 = rng.choice(ACGT, junk_q[0]).tobytes(), rng.choice(ACGT, junk_q[1]).tobytes()
    jt0, jt1 = rng.choice(ACGT, junk_t[0]).tobytes(), rng.choice(ACGT, junk_t[1]).tobytes()
    q, t = jq0 + core_q + jq1, jt0 + core_t + jt1
    mb = [m for m in difflib.SequenceMatcher(None, core_q, core_t, autojunk=False).get_matching_blocks() if m.size >= 11]
    blocks = [(0, 0, 4)] if junk_q[0] >= 4 and junk_t[0] >= 4 else []
    blocks += [(len(jq0) + m.a, len(jt0) + m.b, m.size) for m in mb]
    if junk_q[1] >= 4 and junk_t[1] >= 4:
        blocks.append((len(q) - 4, len(t) - 4, 4))
    return q, t, blocks


def test_banded_local_optimum_is_the_unbanded_one(oracle):
    rng = np.random.default_rng(5)
    for trial in range(12):
        q, t, blocks = _case(rng, int(rng.integers(150, 500)), (int(rng.integers(4, 60)), int(rng.integers(4, 60))),
                             (int(rng.integers(4, 60)), int(rng.integers(4, 60))))
        P = oracle.make_params(align_type=oracle.ALIGN_LOCAL, band=60)
        r, b = oracle.align(q, t, blocks, P)
        assert r["status"] == 0
        assert r["dp_score"] == oracle.full_local_score(q, t, P), trial
        g, _ = oracle.align(q, t, blocks, oracle.make_params(band=60))
        assert r["dp_score"] <= g["dp_score"]                      # free ends never cost
        # the path is consistent with its own counts and lies inside the region
        qs, ts = int(b[0]["q_pos"]), int(b[0]["t_pos"])
        qe, te = int(b[-1]["q_pos"] + b[-1]["length"]), int(b[-1]["t_pos"] + b[-1]["length"])
        assert qe - qs == r["n_match"] + r["n_mismatch"] + r["n_ins"] and te - ts == r["n_match"] + r["n_mismatch"] + r["n_del"]
        assert qe <= r["rows"] and te <= r["cols"]
        # with ins = del = indel the rescored path is the DP optimum
        P2 = oracle.make_params(align_type=oracle.ALIGN_LOCAL, band=60, ins=5, dele=5, indel=5)
        r2, _ = oracle.align(q, t, blocks, P2)
        assert r2["score"] == r2["dp_score"]


def test_local_hand_cases(oracle):
    P = oracle.make_params(align_type=oracle.ALIGN_LOCAL)
    # identical sequences: the whole thing, as the global rule gives
    r, b = oracle.align("ACGTACGTACGT", "ACGTACGTACGT", [(0, 0, 12)], P)
    assert (r["status"], r["dp_score"], r["n_match"], r["n_ins"], r["n_del"]) == (0, -60, 12, 0, 0) and [tuple(x) for x in b] == [(0, 0, 12)]
    # mismatching ends are left off: TT..AA against GG..CC around a common core
    r, b = oracle.align("TTACGTACGTACAA", "GGACGTACGTACCC", [(0, 0, 14)], P)
    assert (r["dp_score"], r["n_match"], r["n_mismatch"]) == (-50, 10, 0) and [tuple(x) for x in b] == [(2, 2, 10)]
    assert (r["q_pos"], r["t_pos"], r["rows"], r["cols"]) == (0, 0, 14, 14)     # the region is still the guide's
    # nothing aligns: every column a mismatch
    r, b = oracle.align("AAAAAAAAAAAA", "CCCCCCCCCCCC", [(0, 0, 12)], P)
    assert r["status"] == oracle.NO_ALIGNMENT and len(b) == 0
    # linear gaps only
    assert oracle.align("ACGT", "ACGT", [(0, 0, 4)], oracle.make_params(align_type=1, mode=oracle.MODE_AFFINE))[0]["status"] != 0


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["lane-per-pair", "wave-per-pair"])
@pytest.mark.parametrize("band,tie", [(12, 0), (12, 1), (90, 0)])
def test_local_rule_on_the_device(oracle, monkeypatch, band, tie, family):
    """Reads whose ends do not align (unrelated sequence in front of and behind a common core, inside the DP region), pairs
    that align end to end and pairs in which nothing aligns; narrow and wide bands; host and device batches; on the lane
    classes (k_fill<false, false, true>, the lane-per-pair traceback from the cheapest cell; band 90: the pairs narrow
    enough for a lane class) and one wave per pair."""
    monkeypatch.setenv("GBA_WAVE_PAIRS", "1" if family == "lane-per-pair" else "4000000000")
    import torch
    from blasr_amd import api, _lib
    rng = np.random.default_rng(40 + band + tie)
    cases = []
    for k in range(60):
        if k % 10 == 9:         # nothing in common: every column a mismatch
            cases.append((b"A" * 120, b"C" * 130, [(0, 0, 4), (116, 126, 4)]))
        elif k % 10 == 8:       # no junk at all
            cases.append(_case(rng, int(rng.integers(200, 900)), (0, 0), (0, 0)))
        else:
            cases.append(_case(rng, int(rng.integers(200, 1500)), (int(rng.integers(4, 120)), int(rng.integers(4, 120))),
                               (int(rng.integers(4, 120)), int(rng.integers(4, 120))), rate=float(rng.choice([0.05, 0.15]))))
    n = len(cases)
    q_len = np.array([len(c[0]) for c in cases], np.uint32); t_len = np.array([len(c[1]) for c in cases], np.uint32)
    n_blk = np.array([len(c[2]) for c in cases], np.uint32)
    off = lambda a: np.concatenate(([0], np.cumsum(a[:-1].astype(np.uint64)))).astype(np.uint64)
    blk = np.zeros(int(n_blk.sum()), _lib.BLOCK_DT)
    x = 0
    for c in cases:
        for b in c[2]:
            blk[x] = b; x += 1
    batch = {"n_pairs": n, "q_arena": np.frombuffer(b"".join(c[0] for c in cases), np.uint8), "t_arena": np.frombuffer(b"".join(c[1] for c in cases), np.uint8),
             "blocks": blk, "q_off": off(q_len), "t_off": off(t_len), "blk_off": off(n_blk), "q_len": q_len, "t_len": t_len, "n_blk": n_blk}
    kw = dict(band=band, tie_break=tie)
    ctx = api.Context(0, timing=True)
    try:
        ctx.set_scoring(api.make_scoring(align_type=_lib.ALIGN_LOCAL, **kw))
        pairs, blocks, rc = ctx.refine_host(batch)
        assert rc == 0
        wide = int(ctx.timing()["wide_pairs"])
        assert wide >= n - 6 if family == "wave-per-pair" else (wide < n // 2 if band == 12 else wide <= n), wide
        OP = oracle.make_params(align_type=oracle.ALIGN_LOCAL, **kw)
        trimmed = empty = 0
        for p, (q, t, b) in enumerate(cases):
            want, wb = oracle.align(q, t, b, OP)
            g = pairs[p]
            assert int(g["status"]) == want["status"], (p, int(g["status"]), want["status"])
            if want["status"] != 0:
                empty += 1
                continue
            for f in util.FIELDS:
                assert int(g[f]) == int(want[f]), (f, p, int(g[f]), int(want[f]))
            kb = blocks[int(g["blk_off"]):int(g["blk_off"]) + int(g["n_blk"])]
            assert (kb == wb).all(), p
            assert (int(g["q_start"]), int(g["q_end"])) == (int(g["q_pos"]) + int(wb[0]["q_pos"]), int(g["q_pos"]) + int(wb[-1]["q_pos"] + wb[-1]["length"]))
            trimmed += int(g["q_end"]) - int(g["q_start"]) < int(g["rows"])
        assert empty == 6 and trimmed >= 40
        # the same batch resident on the device
        d = api.to_device(batch)
        out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
        out_blocks = torch.zeros((len(blocks) + 8, 3), dtype=torch.int32, device="cuda:0")
        rc, needed = ctx.refine_device(d, out_pairs, out_blocks)
        torch.cuda.synchronize()
        assert rc == 0 and out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT).tobytes() == pairs.tobytes()
        # the global rule on the same context afterwards is untouched by it
        ctx.set_scoring(api.make_scoring(**kw))
        gp, gb, rc = ctx.refine_host(batch)
        util_ok = 0
        for p, (q, t, b) in enumerate(cases):
            want, wb = oracle.align(q, t, b, oracle.make_params(**kw))
            if want["status"] == 0:
                assert int(gp[p]["dp_score"]) == want["dp_score"] and int(gp[p]["score"]) == want["score"]
                util_ok += 1
        assert util_ok >= 50
        # affine gaps with the local rule are refused, the scoring in force stays
        with pytest.raises(_lib.GbaError):
            ctx.set_scoring(api.make_scoring(align_type=_lib.ALIGN_LOCAL, mode=_lib.MODE_AFFINE, band=16))
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["lane-per-pair", "wave-per-pair"])
def test_local_rule_with_a_cost_profile(oracle, monkeypatch, family):
    """The CCS call site runs the Local rule with the IDS scorer: a per-base cost profile through the same kernels, one
    lane per pair (k_fill<false, true, true>) and one wave per pair."""
    from blasr_amd import api, _lib
    monkeypatch.setenv("GBA_WAVE_PAIRS", "1" if family == "lane-per-pair" else "4000000000")
    rng = np.random.default_rng(77)
    cases = [_case(rng, int(rng.integers(200, 700)), (int(rng.integers(4, 80)), int(rng.integers(4, 80))),
                   (int(rng.integers(4, 80)), int(rng.integers(4, 80)))) for _ in range(24)]
    n = len(cases)
    q_len = np.array([len(c[0]) for c in cases], np.uint32); t_len = np.array([len(c[1]) for c in cases], np.uint32)
    n_blk = np.array([len(c[2]) for c in cases], np.uint32)
    off = lambda a: np.concatenate(([0], np.cumsum(a[:-1].astype(np.uint64)))).astype(np.uint64)
    blk = np.array([b for c in cases for b in c[2]], dtype=_lib.BLOCK_DT)
    qa = np.frombuffer(b"".join(c[0] for c in cases), np.uint8)
    prof = np.zeros((len(qa), 11), np.int16)
    prof[:, :5] = rng.integers(4, 20, (len(qa), 5))
    code = np.full(256, 4, np.int64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    prof[np.arange(len(qa)), code[qa]] = rng.integers(-7, -2, len(qa))
    prof[:, 5] = rng.integers(2, 12, len(qa)); prof[:, 6:] = rng.integers(2, 12, (len(qa), 5))
    prof = prof.reshape(-1)
    batch = {"n_pairs": n, "q_arena": qa, "t_arena": np.frombuffer(b"".join(c[1] for c in cases), np.uint8), "blocks": blk,
             "q_off": off(q_len), "t_off": off(t_len), "blk_off": off(n_blk), "q_len": q_len, "t_len": t_len, "n_blk": n_blk, "q_profile": prof}
    ctx = api.Context(0, timing=True)
    try:
        ctx.set_scoring(api.make_scoring(align_type=_lib.ALIGN_LOCAL, band=12))
        pairs, blocks, rc = ctx.refine_host(batch)
        assert rc == 0
        wide = int(ctx.timing()["wide_pairs"])
        assert wide == n if family == "wave-per-pair" else wide < n // 2, wide
        OP = oracle.make_params(align_type=oracle.ALIGN_LOCAL, band=12)
        for p, (q, t, b) in enumerate(cases):
            o = int(batch["q_off"][p])
            want, wb = oracle.align_qv(q, t, b, OP, prof[o * 11:(o + len(q)) * 11])
            g = pairs[p]
            assert int(g["status"]) == want["status"] == 0, p
            for f in util.FIELDS:
                assert int(g[f]) == int(want[f]), (f, p)
            assert (blocks[int(g["blk_off"]):int(g["blk_off"]) + int(g["n_blk"])] == wb).all(), p
    finally:
        ctx.close()
