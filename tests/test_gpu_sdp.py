"""Guide production on the GPU (gba_sdp_device, through the C-ABI) against its CPU restatement
oracle/sdp_oracle.c, bit for bit, and the guides it makes fed to the refine stage.  Parity with
libcpp's SDPAlign is unpinned (see the oracle's header)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from blasr_amd import api as a
    return a


def _guides_to_host(g):
    from blasr_amd import _lib
    st = g["status"].cpu().numpy()
    sc = g["chain_score"].cpu().numpy()
    nb = g["n_blk"].cpu().numpy().view(np.uint32)
    off = g["blk_off"].cpu().numpy().view(np.uint64)
    blocks = np.frombuffer(g["blocks"].cpu().numpy().tobytes(), dtype=_lib.BLOCK_DT)
    return st, sc, nb, off, blocks


def _compare(api, oracle, ctx, batch, dbatch=None, **kw):
    import torch
    P = api.make_sdp_params(**kw)
    OP = oracle.make_sdp_params(k=P.tuple_size, ins=P.ins, dele=P.dele, match=P.match, max_fragments=P.max_fragments)
    g = ctx.sdp_device(dbatch if dbatch is not None else api.to_device(batch), P)
    torch.cuda.synchronize()
    assert g["rc"] == 0
    st, sc, nb, off, blocks = _guides_to_host(g)
    est, enb, esc, _, eb = oracle.sdp_chain_batch(batch, OP)
    assert (st == est).all(), np.flatnonzero(st != est)[:10]
    assert (nb == enb).all() and (sc == esc).all()
    n = int(batch["n_pairs"])
    assert int(g["blocks_needed"]) == int(nb.sum())
    want_off = np.zeros(n, np.uint64); want_off[1:] = np.cumsum(nb[:-1].astype(np.uint64))
    assert (off == want_off).all()
    for p in range(n):
        got = blocks[int(off[p]):int(off[p]) + int(nb[p])]
        assert (got == eb[p]).all(), p
    return st, g


@pytest.mark.parametrize("tile", ["1024", "8192"])
def test_parity_with_the_oracle(api, oracle, synth, tile, monkeypatch):
    monkeypatch.setenv("GBA_SDP_TILE", tile)
    ctx = api.Context(0)
    try:
        st, _ = _compare(api, oracle, ctx, synth.generate(48, read_len=10_000, seed=31))
        assert (st == 0).all()
        st, _ = _compare(api, oracle, ctx, synth.generate(300, read_len=1200, seed=32, len_jitter=900, n_frac=0.02))
        assert (st == 0).sum() > 280
        # decoys: unrelated windows give short chains of chance matches or nothing at all
        st, _ = _compare(api, oracle, ctx, synth.generate(120, read_len=3000, seed=33, decoy_every=4, share_reads=True))
        # a short tuple and a small fragment budget: many pairs are too repetitive for it
        st, _ = _compare(api, oracle, ctx, synth.generate(80, read_len=2000, seed=34, len_jitter=1000), tuple_size=6, max_fragments=256)
        assert (st == 8).any() and (st == 0).any()
        st, _ = _compare(api, oracle, ctx, synth.generate(40, read_len=2500, seed=35), tuple_size=15, ins=3, dele=3, match=7, max_fragments=200)
    finally:
        ctx.close()


def test_edge_cases(api, oracle):
    import util
    ctx = api.Context(0)
    try:
        rng = np.random.default_rng(5)
        rnd = lambda n: bytes(rng.choice(list(b"ACGT"), n).tolist())
        a = rnd(500)
        pairs = [(b"ACGT", a, []), (a, b"ACG", []), (a, a, []), (a, a[::-1], []),
                 (b"A" * 400, b"A" * 400, []), (b"N" * 64, b"N" * 64, []), (a[:100] + b"N" + a[100:], a, []),
                 (a.lower(), a, []), (a[:250] + a[:250] + a[250:], a, []), (a, a[:250] + a[200:], []),
                 (b"ACGTTGCAAGG", b"ACGTTGCAAGG", []), (rnd(3000), rnd(3000), [])]
        batch = util.make_batch([(q, t, [(0, 0, 1)]) for q, t, _ in pairs])
        st, _ = _compare(api, oracle, ctx, batch, max_fragments=128)
        assert [int(s) for s in st[:3]] == [1, 1, 0] and [int(s) for s in st[4:6]] == [8, 1] and int(st[10]) == 0
    finally:
        ctx.close()


def test_windows_of_a_resident_reference_on_both_strands(api, oracle, synth):
    ctx = api.Context(0)
    try:
        ref = synth.reference(3_000_000, n_frac=0.01)
        batch = synth.generate(200, read_len=2500, seed=36, ref=ref, len_jitter=1000)
        assert 50 < int(batch["win_rc"].sum()) < 150
        ctx.set_reference(ref)
        st, g = _compare(api, oracle, ctx, batch, dbatch=api.to_device(synth.as_windows(batch)))
        assert (st == 0).sum() > 190
        # the same through the host-buffer entry point
        hst, hsc, hnb, hoff, hblocks, rc = ctx.sdp_host(synth.as_windows(batch), api.make_sdp_params())
        dst, dsc, dnb, doff, dblocks = _guides_to_host(g)
        assert rc == 0 and (hst == dst).all() and (hsc == dsc).all() and (hnb == dnb).all() and (hoff == doff).all()
        assert (hblocks == dblocks[:len(hblocks)]).all() and len(hblocks) == int(dnb.sum())
    finally:
        ctx.close()


def test_guides_feed_the_refine_stage(api, oracle, synth):
    """Reads -> guides on the device -> refine on the device, nothing travels in between.  The
    refined alignments are compared with the ones the truth-derived guides of the generator give."""
    import torch
    from blasr_amd import _lib
    ctx = api.Context(0)
    try:
        batch = synth.generate(256, read_len=8000, seed=37)
        n = 256
        db = api.to_device(batch)
        ctx.set_scoring(api.make_scoring())
        out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
        out_blocks = torch.zeros((int(batch["q_len"].sum()) + n, 3), dtype=torch.int32, device="cuda:0")
        rc, _ = ctx.refine_device(db, out_pairs, out_blocks)
        assert rc == 0
        truth = np.frombuffer(out_pairs.cpu().numpy().tobytes(), dtype=_lib.PAIR_RESULT_DT).copy()
        g = ctx.sdp_device(db, api.make_sdp_params())
        assert g["rc"] == 0 and int((g["status"] != 0).sum()) == 0
        db2 = dict(db, blocks=g["blocks"], blk_off=g["blk_off"], n_blk=g["n_blk"])
        rc, _ = ctx.refine_device(db2, out_pairs, out_blocks)
        torch.cuda.synchronize()
        assert rc == 0
        mine = np.frombuffer(out_pairs.cpu().numpy().tobytes(), dtype=_lib.PAIR_RESULT_DT)
        assert (mine["status"] == 0).all() and (truth["status"] == 0).all()
        # the chain starts at the first and ends at the last anchor, so the refined stretch is a little
        # shorter than with the generator's guide; per aligned base the alignments are as good
        per_base_mine = mine["score"] / mine["rows"]
        per_base_truth = truth["score"] / truth["rows"]
        assert (mine["rows"] >= 0.97 * truth["rows"]).all()
        assert np.mean(per_base_mine <= per_base_truth + 0.02) > 0.97
        # the refined alignments agree with the oracle run on the same guides
        st, sc, nb, off, blocks = _guides_to_host(g)
        hb = dict(batch, blocks=blocks, blk_off=off, n_blk=nb)
        res, _ = oracle.align_batch(hb, oracle.make_params(), first=0, count=16)
        for f in ("score", "n_match", "n_mismatch", "n_ins", "n_del"):
            assert (res[f] == mine[f][:16]).all(), f
    finally:
        ctx.close()


def test_host_mirror_guides_then_refine(api, oracle, synth):
    """The read-level mirror: SDPAlignCandidates fills the candidates' guides, RefineAlignments refines
    them; both against the oracle's two stages run one after the other on the same sequences."""
    from blasr_amd import refine
    batch = synth.generate(6, read_len=2500, seed=41)
    ctx = api.Context(0)
    try:
        params = refine.MappingParameters()
        read = bytes(synth.pair(batch, 0)[0])
        # the read's own window, a window of another read (chance matches at best), and its own window cut short
        windows = [bytes(synth.pair(batch, 0)[1]), bytes(synth.pair(batch, 1)[1]), bytes(synth.pair(batch, 0)[1])[300:1800]]
        cands = [refine.AlignmentCandidate(tAlignedSeq=w) for w in windows]
        refine.SDPAlignCandidates(read, cands, params, ctx)
        op = oracle.make_params()
        expect = []
        for c, w in zip(cands, windows):
            st, blks, score, _ = oracle.sdp_chain(read, w, oracle.make_sdp_params(max_fragments=1024))
            assert c.status == st
            if st != 0:
                assert len(c.blocks) == 0
                continue
            assert c.score == -score and (c.qPos, c.tPos) == (int(blks[0]["q_pos"]), int(blks[0]["t_pos"]))
            assert (c.blocks["q_pos"] + c.qPos == blks["q_pos"]).all() and (c.blocks["t_pos"] + c.tPos == blks["t_pos"]).all()
            assert (c.blocks["length"] == blks["length"]).all()
            # what the refine stage must make of this guide: the oracle on the slices RefineAlignment cuts (:693-703)
            last = c.blocks[-1]
            ql, tl = int(last["q_pos"]) + int(last["length"]), int(last["t_pos"]) + int(last["length"])
            res, rb = oracle.align(read[c.qPos:c.qPos + ql], w[c.tPos:c.tPos + tl], c.blocks, op)
            expect.append((w, res, rb))
        kept = [c for c in cands if c.status == 0]
        refine.RefineAlignments(read, kept, params, ctx)
        assert kept[0].nMatch > 1500 and kept[0].score <= kept[-1].score         # sorted by score, the true locus first
        for w, res, rb in expect:
            c = next(k for k in kept if k.score == res["score"] and k.nMatch == res["n_match"])
            assert (c.nMismatch, c.nIns, c.nDel) == (res["n_mismatch"], res["n_ins"], res["n_del"])
            assert len(c.blocks) == len(rb) and (c.blocks["length"] == rb["length"]).all()
    finally:
        ctx.close()


def test_recorded_fuzz_mismatches(api, oracle):
    """Every pair on which a randomised run once disagreed with the oracle, replayed."""
    import json, os, util
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sdp_fuzz_regressions.json")))["cases"]
    ctx = api.Context(0)
    try:
        for c in cases:
            batch = util.make_batch([(c["q"].encode("latin1"), c["t"].encode("latin1"), [(0, 0, 1)])] * 3)
            _compare(api, oracle, ctx, batch, **c["sdp"])
    finally:
        ctx.close()


def test_block_arena_too_small_and_long_reads(api, oracle, synth):
    """NOSPACE: the count is still reported, pairs that fit are complete, the others are flagged.  And a
    batch of 50-kb reads with a fragment budget that fits them (the default 1024 is for reads up to ~30 kb)."""
    import torch
    from blasr_amd import _lib
    ctx = api.Context(0)
    try:
        batch = synth.generate(40, read_len=3000, seed=51)
        db = api.to_device(batch)
        full = ctx.sdp_device(db, api.make_sdp_params())
        st, sc, nb, off, blocks = _guides_to_host(full)
        assert full["rc"] == 0 and (st == 0).all()
        cap = int(off[25])
        g = ctx.sdp_device(db, api.make_sdp_params(), blocks_capacity=cap)
        torch.cuda.synchronize()
        assert g["rc"] == _lib.E_NOSPACE and g["blocks_needed"] == full["blocks_needed"]
        st2, _, nb2, off2, blocks2 = _guides_to_host(g)
        assert (st2[:25] == 0).all() and (st2[25:] == _lib.PAIR_NOSPACE).all() and (off2 == off).all() and (nb2 == nb).all()
        assert (blocks2[:cap] == blocks[:cap]).all()
        long_batch, _ = synth.preset("c5_50kb_wide", 6)
        st, _ = _compare(api, oracle, ctx, long_batch, max_fragments=4096)
        assert (st == 0).all()
        st, _ = _compare(api, oracle, ctx, long_batch, max_fragments=512)
        assert (st == _lib.PAIR_TOO_MANY).all()
    finally:
        ctx.close()


def test_guided_align_from_sequences(api, oracle, synth):
    """The composite the CCS path uses (chain of 6-mers, then a band of 12 around it with the Local end rule) against the two
    oracles run one after the other; reads with a stretch of unrelated sequence at one end, which the local rule leaves
    off; the C++ mirror gives the same."""
    import os
    import subprocess
    from blasr_amd import build, refine
    batch = synth.generate(4, read_len=1200, seed=61, p_ins=0.02, p_del=0.02, p_sub=0.01)
    rng = np.random.default_rng(3)
    ctx = api.Context(0)
    try:
        params = refine.MappingParameters(sortRefinedAlignments=False, sdpMaxFragments=4096)
        lines, got = ["4"], []
        for p in range(4):
            q, t, _ = synth.pair(batch, p)
            read, window = bytes(q), bytes(t)
            # a 6-mer of the window repeated inside an unrelated read tail: the chain reaches into the tail, the alignment should not
            tail = rng.choice(np.frombuffer(b"ACGT", np.uint8), 90).tobytes()
            read = read + tail[:40] + window[-6:] + tail[40:]
            cands = [refine.AlignmentCandidate(tAlignedSeq=window)]
            refine.GuidedAlignFromSequences(read, cands, params, ctx)
            st, guide, _, _ = oracle.sdp_chain(read, window, oracle.make_sdp_params(k=6, max_fragments=4096))
            assert st == 0 and cands[0].status == 0
            q0, t0 = int(guide[0]["q_pos"]), int(guide[0]["t_pos"])
            g = guide.copy(); g["q_pos"] -= np.uint32(q0); g["t_pos"] -= np.uint32(t0)
            ql, tl = int(g[-1]["q_pos"]) + int(g[-1]["length"]), int(g[-1]["t_pos"]) + int(g[-1]["length"])
            er, eb = oracle.align(read[q0:q0 + ql], window[t0:t0 + tl], g, oracle.make_params(band=12, align_type=oracle.ALIGN_LOCAL))
            c = cands[0]
            assert (c.score, c.nMatch, c.nMismatch, c.nIns, c.nDel) == (er["score"], er["n_match"], er["n_mismatch"], er["n_ins"], er["n_del"])
            assert len(c.blocks) == len(eb) and (c.blocks == eb).all()
            assert c.nMatch > 0.9 * len(q)
            assert c.qInterval[1] <= len(q) + 8          # the unrelated tail is not part of the alignment
            lines.append("%s %s 0" % (read.decode(), window.decode()))
            got.append(c)
        exe = build.HOST_DEMO
        assert os.path.exists(exe), "run python -m blasr_amd.build"
        out = subprocess.run([exe, "ccs"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        for row, c in zip(out.stdout.strip().split("\n"), got):
            v = [int(x) for x in row.split()]
            assert v[:6] == [c.status, c.score, c.nCells, c.qPos, c.tPos, len(c.blocks)]
            assert v[6:] == [int(x) for b in c.blocks for x in b]
    finally:
        ctx.close()
