"""Real PacBio subreads with their per-base quality values (tests/golden/pacbio_subreads_qv.json, made by
tests/golden/make_pacbio_qv_fixture.py from the subreads BAM under the reference's utils/bam2bax/tests/data):
realistic cost profiles for the quality-value scorers.  CPU: the oracles reproduce the stored results.  GPU: the
guided DP with a profile (linear and affine gaps) and gba_kband give the oracle's results bit for bit."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pacbio_subreads_qv.json")
FIELDS = ("status", "dp_score", "score", "q_pos", "t_pos", "n_blk", "n_match", "n_mismatch", "n_ins", "n_del", "rows", "cols", "ncells")


def _cases():
    return json.load(open(GOLD))["cases"]


def _qv(refine, c):
    dec = lambda s: bytes(ord(ch) - 33 for ch in s)
    return refine.QualityValues(qual=None, insertionQV=dec(c["iq"]), deletionQV=dec(c["dq"]), deletionTag=c["dt"].encode(),
                                substitutionQV=dec(c["sq"]), substitutionTag=None)


def _slices(refine, c, params):
    blocks = np.array([tuple(b) for b in c["blocks"]], dtype=refine.BLOCK_DT)
    ql, tl = int(blocks[-1]["q_pos"] + blocks[-1]["length"]), int(blocks[-1]["t_pos"] + blocks[-1]["length"])
    q0, t0 = c["q0"], c["t0"]
    prof = refine.ids_profile(c["read"].encode(), _qv(refine, c), params)
    return c["read"][q0:q0 + ql].encode(), c["target"][t0:t0 + tl].encode(), blocks, prof[q0 * 11:(q0 + ql) * 11]


def _params(oracle, refine, params):
    lin = oracle.make_params(ins=params.deletion, dele=params.insertion, indel=params.indel, band=params.guidedAlignBandSize)
    aff = oracle.make_params(ins=params.deletion, dele=params.insertion, indel=params.indel, band=params.bandSize, mode=oracle.MODE_AFFINE,
                             affine_open=params.affineOpen, affine_extend=params.affineExtend, stats_affine=1)
    return lin, aff


def test_quality_values_are_real_and_inside_the_cost_ranges():
    from blasr_amd import refine
    params = refine.MappingParameters(ignoreQualities=False)
    seen = set()
    for c in _cases():
        assert len(c["read"]) == len(c["iq"]) == len(c["dq"]) == len(c["dt"]) == len(c["sq"])
        prof = refine.ids_profile(c["read"].encode(), _qv(refine, c), params).reshape(-1, 11)
        assert int(np.abs(prof).max()) <= 63          # real quality values stay inside the affine kernels' +-63 (and the +-127 of the linear ones)
        seen |= set(np.unique(prof[:, 5]).tolist())
    assert len(seen) > 10                              # insertion costs really vary from base to base


def test_oracles_reproduce_the_stored_results(oracle):
    from blasr_amd import api, refine, _lib
    params = refine.MappingParameters(ignoreQualities=False)
    lin, aff = _params(oracle, refine, params)
    KP = api.make_kband_params(align_type=_lib.KB_GLOBAL, ins=params.insertion, dele=params.deletion)
    for c in _cases():
        qs, ts, blocks, prof = _slices(refine, c, params)
        for key, P in (("guided_linear", lin), ("guided_affine", aff)):
            res, blk = oracle.align_qv(qs, ts, blocks, P, prof)
            assert {a: int(b) for a, b in res.items()} == c["expect"][key]["result"], (c["name"], key)
            assert [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in blk] == c["expect"][key]["blocks"]
        e = c["expect"]["kband_global"]
        res, blk = oracle.kband_align(qs, ts, e["k"], KP, profile=prof)
        assert {a: int(b) for a, b in res.items()} == e["result"], c["name"]
        assert [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in blk] == e["blocks"]


@pytest.mark.gpu
@pytest.mark.parametrize("affine", [False, True])
def test_guided_dp_with_real_quality_values(affine):
    """gba_submit with gba_batch.q_profile built from the real quality values, against the stored oracle results."""
    from blasr_amd import api, refine, _lib
    params = refine.MappingParameters(ignoreQualities=False, affineAlign=affine)
    cases = _cases()
    parts = [_slices(refine, c, params) for c in cases]
    n = len(parts)
    q_len = np.array([len(p[0]) for p in parts], np.uint32); t_len = np.array([len(p[1]) for p in parts], np.uint32)
    n_blk = np.array([len(p[2]) for p in parts], np.uint32)
    off = lambda a: np.concatenate(([0], np.cumsum(a[:-1].astype(np.uint64)))).astype(np.uint64)
    batch = {"n_pairs": n, "q_arena": np.frombuffer(b"".join(p[0] for p in parts), np.uint8), "t_arena": np.frombuffer(b"".join(p[1] for p in parts), np.uint8),
             "blocks": np.concatenate([p[2] for p in parts]), "q_off": off(q_len), "t_off": off(t_len), "blk_off": off(n_blk),
             "q_len": q_len, "t_len": t_len, "n_blk": n_blk, "q_profile": np.concatenate([p[3] for p in parts])}
    ctx = api.Context(0)
    try:
        ctx.set_scoring(refine.scoring_from_params(params, ids=True))
        pairs, blocks, rc = ctx.refine_host(batch)
        assert rc == 0
        key = "guided_affine" if affine else "guided_linear"
        for i, c in enumerate(cases):
            e = c["expect"][key]
            for f in FIELDS:
                assert int(pairs[i][f]) == e["result"][f], (c["name"], f)
            got = blocks[int(pairs[i]["blk_off"]):int(pairs[i]["blk_off"]) + int(pairs[i]["n_blk"])]
            assert [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in got] == e["blocks"], c["name"]
    finally:
        ctx.close()


@pytest.mark.gpu
def test_kband_with_real_quality_values():
    """gba_kband (KBandAlign with the IDS scorer, Global) on the same reads and costs, against the stored oracle results."""
    from blasr_amd import api, refine, _lib
    from kband_cases import pack
    params = refine.MappingParameters(ignoreQualities=False)
    cases = _cases()
    parts = [_slices(refine, c, params) for c in cases]
    pairs_in = [(np.frombuffer(p[0], np.uint8), np.frombuffer(p[1], np.uint8)) for p in parts]
    b = pack(pairs_in, [c["expect"]["kband_global"]["k"] for c in cases], profile=np.concatenate([p[3] for p in parts]))
    ctx = api.Context(0)
    try:
        got, blk, rc = ctx.kband_host(b, api.make_kband_params(align_type=_lib.KB_GLOBAL, ins=params.insertion, dele=params.deletion))
        assert rc == 0
        for i, c in enumerate(cases):
            e = c["expect"]["kband_global"]
            for f in FIELDS:
                assert int(got[i][f]) == e["result"][f], (c["name"], f)
            mine = blk[int(got[i]["blk_off"]):int(got[i]["blk_off"]) + int(got[i]["n_blk"])]
            assert [[int(x["q_pos"]), int(x["t_pos"]), int(x["length"])] for x in mine] == e["blocks"], c["name"]
    finally:
        ctx.close()
