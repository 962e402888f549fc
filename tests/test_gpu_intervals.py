"""The aligned interval of the pair results (q_start .. t_end: GetQInterval and the -m 4 columns
qStart qEnd tStart tEnd) and batches run without a block arena (gba_results.blocks == NULL: interval
and summary output), through the C-ABI, against the oracle's blocks and against the same batch run
with an arena."""
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SUMMARY_FIELDS = ["status", "dp_score", "score", "q_pos", "t_pos", "n_blk", "blk_off", "n_match", "n_mismatch", "n_ins", "n_del",
                  "rows", "cols", "ncells", "extra_score", "q_start", "q_end", "t_start", "t_end"]


@pytest.fixture(scope="module")
def api():
    from blasr_amd import api as a
    return a


def _interval_of(r, blocks):
    """(q_start, q_end, t_start, t_end) from a pair's position and its blocks."""
    qp, tp = int(r["q_pos"]), int(r["t_pos"])
    if len(blocks) == 0:
        return qp, qp, tp, tp
    a, z = blocks[0], blocks[-1]
    return (qp + int(a["q_pos"]), qp + int(z["q_pos"]) + int(z["length"]),
            tp + int(a["t_pos"]), tp + int(z["t_pos"]) + int(z["length"]))


def _same_summary(a, b):
    for f in SUMMARY_FIELDS:
        assert np.array_equal(a[f], b[f]), f


@pytest.mark.parametrize("family", ["lane-per-pair", "wave-per-pair"])
def test_intervals_match_the_oracle_and_need_no_blocks(api, oracle, synth, family, monkeypatch):
    from blasr_amd import _lib
    v = "0" if family == "lane-per-pair" else "4000000000"
    monkeypatch.setenv("GBA_WAVE_PAIRS", v)
    monkeypatch.setenv("GBA_WAVE_PAIRS_AFFINE", v)
    ctx = api.Context(0)
    try:
        batch = synth.generate(120, read_len=1200, seed=909, len_jitter=600, n_frac=0.01)
        # one pair with nothing to align, one whose guide is broken
        batch["n_blk"][7] = 0
        batch["blk_off"][11] = np.uint64(len(batch["blocks"]) + 5)
        for kw, okw in ((dict(), dict()),
                        (dict(mode=_lib.MODE_AFFINE, band=16, affine_open=10, affine_extend=1),
                         dict(mode=1, band=16, affine_open=10, affine_extend=1))):
            ctx.set_scoring(api.make_scoring(**kw))
            full, blocks, rc = ctx.refine_host(batch)
            assert rc == 0
            only, rc2 = ctx.refine_host_intervals(batch)
            assert rc2 == 0 and ctx.blocks_needed == len(blocks)
            _same_summary(full, only)
            assert int(full["status"][7]) != 0 and int(full["status"][11]) != 0
            P = oracle.make_params(**okw)
            n_ok = 0
            for p in range(int(batch["n_pairs"])):
                r = full[p]
                if int(r["status"]) != 0:
                    assert (int(r["q_start"]), int(r["q_end"]), int(r["t_start"]), int(r["t_end"])) == (0, 0, 0, 0)
                    continue
                n_ok += 1
                kb = blocks[int(r["blk_off"]):int(r["blk_off"]) + int(r["n_blk"])]
                got = tuple(int(r[f]) for f in ("q_start", "q_end", "t_start", "t_end"))
                assert got == _interval_of(r, kb), p
                if p % 6 == 0:
                    q, t, g = synth.pair(batch, p)
                    er, eb = oracle.align(q, t, g, P)
                    assert got == _interval_of(er, eb), p
            assert n_ok >= int(batch["n_pairs"]) - 2
    finally:
        ctx.close()


def test_an_all_gap_path_has_an_empty_interval(api, oracle):
    """No column of the path is gap-free (mismatches cost more than an insertion plus a deletion)."""
    from blasr_amd.synth import BLOCK_DT
    ctx = api.Context(0)
    try:
        ctx.set_scoring(api.make_scoring(match=-1, mismatch=50, ins=1, dele=1, indel=1))
        q, t = b"AAAA", b"CCCC"
        g = np.array([(0, 0, 4)], dtype=BLOCK_DT)
        batch = {"n_pairs": 1, "q_arena": np.frombuffer(q, np.uint8), "q_off": np.zeros(1, np.uint64), "q_len": np.array([4], np.uint32),
                 "t_arena": np.frombuffer(t, np.uint8), "t_off": np.zeros(1, np.uint64), "t_len": np.array([4], np.uint32),
                 "blocks": g, "blk_off": np.zeros(1, np.uint64), "n_blk": np.array([1], np.uint32)}
        er, eb = oracle.align(q, t, g, oracle.make_params(match=-1, mismatch=50, ins=1, dele=1, indel=1))
        for pairs in (ctx.refine_host(batch)[0], ctx.refine_host_intervals(batch)[0]):
            r = pairs[0]
            assert int(r["status"]) == 0 and int(r["n_blk"]) == er["n_blk"] == 0 and int(r["score"]) == er["score"]
            assert (int(r["q_start"]), int(r["q_end"]), int(r["t_start"]), int(r["t_end"])) == _interval_of(r, [])
    finally:
        ctx.close()


def test_device_entry_points_without_a_block_arena(api, synth):
    """gba_submit_device and its no-allocation variant with blocks == NULL, in several parts, alone and with
    CIGARs: the pair results and the operations are those of the run with an arena, nothing is NOSPACE."""
    import torch
    from blasr_amd import _lib
    ctx = api.Context(0)
    try:
        ctx.set_scoring(api.make_scoring())
        batch = synth.generate(3000, read_len=500, seed=4321, len_jitter=250)
        n = int(batch["n_pairs"])
        db = api.to_device(batch)
        isz = _lib.PAIR_RESULT_DT.itemsize
        cap = int(batch["q_len"].sum()) + n

        def run(ctx, want_blocks, cigar, no_alloc=False):
            pairs = torch.zeros(n * isz, dtype=torch.uint8, device="cuda:0")
            blocks = torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0") if want_blocks else None
            cg = None
            if cigar:
                cg = (torch.zeros(2 * cap, dtype=torch.int32, device="cuda:0"), torch.zeros(n + 1, dtype=torch.int64, device="cuda:0"), 1)
            rc, needed = ctx.refine_device(db, pairs, blocks, no_alloc=no_alloc, cigar=cg)
            assert rc == 0
            res = np.frombuffer(pairs.cpu().numpy().tobytes(), dtype=_lib.PAIR_RESULT_DT)
            ops = None
            if cigar:
                off = cg[1].cpu().numpy()
                ops = cg[0].cpu().numpy()[:int(off[n])]
            return res, needed, ops

        import os
        old = os.environ.get("GBA_PART_PAIRS")
        os.environ["GBA_PART_PAIRS"] = "1024"          # three parts
        try:
            c2 = api.Context(0)
            c2.set_scoring(api.make_scoring())
            full, needed, ops = run(c2, True, True)
            only, needed2, ops2 = run(c2, False, True)
            bare, needed3, _ = run(c2, False, False)
            c2.close()
        finally:
            if old is None:
                os.environ.pop("GBA_PART_PAIRS")
            else:
                os.environ["GBA_PART_PAIRS"] = old
        assert needed == needed2 == needed3 and (full["status"] == 0).all()
        _same_summary(full, only)
        _same_summary(full, bare)
        assert np.array_equal(ops, ops2)
        fixed, needed4, _ = run(ctx, False, False, no_alloc=True)
        _same_summary(full, fixed)
        assert needed4 == needed
    finally:
        ctx.close()


def test_cpp_shim_prints_intervals_without_blocks(oracle, synth):
    from blasr_amd import build
    batch = synth.generate(10, read_len=600, len_jitter=200, seed=1234)
    lines = ["10"]
    for i in range(10):
        q, t, blk = synth.pair(batch, i)
        lines.append("%s %s %d %s" % (bytes(q).decode(), bytes(t).decode(), len(blk),
                                      " ".join("%d %d %d" % (b["q_pos"], b["t_pos"], b["length"]) for b in blk)))
    out = subprocess.run([build.HOST_DEMO, "intervals"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = out.stdout.strip().split("\n")
    assert len(rows) == 10
    for i, row in enumerate(rows):
        v = [int(x) for x in row.split()]
        q, t, blk = synth.pair(batch, i)
        er, eb = oracle.align(q, t, blk, oracle.make_params())
        # the demo's candidates start at read / window position 0, so the printed interval is the pair's
        assert v == [0, er["score"], er["n_blk"], *_interval_of(er, eb)]


def test_mapping_qualities_from_intervals_alone(api, synth):
    """StoreMapQVs needs the read interval of every candidate: the same qualities come out of candidates
    refined with and without a block arena."""
    from blasr_amd import mapqv, refine
    ctx = api.Context(0)
    try:
        batch = synth.generate(6, read_len=900, seed=31)
        q0, t0, g0 = synth.pair(batch, 0)
        read = bytes(q0)
        # six candidates of one read: its own window whole and cut short, and windows of other reads
        def cands():
            out = []
            for i in range(6):
                g = g0[:len(g0) - 2 * i] if i < 3 else g0[:max(1, len(g0) // (i + 1))]
                out.append(refine.AlignmentCandidate(tAlignedSeq=bytes(t0), blocks=g.copy()))
            return out
        params = refine.MappingParameters(sortRefinedAlignments=False)
        a = refine.RefineAlignments(read, cands(), params, ctx)
        b = refine.RefineAlignments(read, cands(), params, ctx, want_blocks=False)
        raw = [c.score for c in a]
        assert [c.score for c in b] == raw and all(len(c.blocks) == 0 and c.nBlocks == len(x.blocks) for c, x in zip(b, a))
        assert [mapqv.q_interval(c) for c in a] == [mapqv.q_interval(c) for c in b] == [c.qInterval for c in a]
        mapqv.store_map_qvs(a, raw)
        mapqv.store_map_qvs(b, raw)
        assert [c.mapQV for c in a] == [c.mapQV for c in b]
        from blasr_amd import emit
        for x, y in zip(a, b):
            assert emit.m4_line(x, "read/0", "chr", len(read), 5000, x.mapQV) == emit.m4_line(y, "read/0", "chr", len(read), 5000, y.mapQV)
        assert len(refine.RemoveLowQualityAlignments(b, params)) == len(refine.RemoveLowQualityAlignments(a, params))
    finally:
        ctx.close()
