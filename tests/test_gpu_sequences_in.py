"""From sequences to refined alignments in one call (gba_set_guide_producer + a batch without guide
blocks), through the C-ABI: the results must be those of the two stages called one after the other
(gba_sdp, then gba_submit on its guides), which the other tests check against the two oracles; a few
pairs are checked against the oracles directly as well."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RESULT_FIELDS = ["dp_score", "score", "q_pos", "t_pos", "n_blk", "blk_off", "n_match", "n_mismatch", "n_ins", "n_del",
                 "rows", "cols", "ncells", "extra_score", "q_start", "q_end", "t_start", "t_end"]
SEQ_KEYS = ("n_pairs", "q_arena", "q_off", "q_len", "t_arena", "t_off", "t_len", "t_rc")


@pytest.fixture(scope="module")
def api():
    from blasr_amd import api as a
    return a


def _sequences_only(batch):
    return {k: batch[k] for k in SEQ_KEYS if batch.get(k) is not None}


def _two_calls(ctx, seqs, sp):
    status, score, n_blk, off, blocks, rc = ctx.sdp_host(seqs, sp)
    assert rc == 0
    guided = dict(seqs, blocks=blocks if len(blocks) else np.zeros(1, blocks.dtype), blk_off=off, n_blk=n_blk)
    pairs, out_blocks, rc = ctx.refine_host(guided)
    assert rc == 0
    return status, pairs, out_blocks


def _same(status, want, want_blocks, got, got_blocks):
    ok = status == 0
    # a pair without a guide carries the producer's reason, every other pair is the two-call result
    assert np.array_equal(got["status"][~ok], status[~ok])
    assert np.array_equal(got["status"][ok], want["status"][ok])
    for f in RESULT_FIELDS:
        assert np.array_equal(got[f][ok], want[f][ok]), f
    assert np.array_equal(got_blocks, want_blocks)


def test_one_call_equals_two_calls(api, oracle, synth):
    from blasr_amd import _lib
    ctx = api.Context(0)
    try:
        batch = synth.generate(200, read_len=2500, seed=1717, len_jitter=1500, n_frac=0.01)
        seqs = _sequences_only(batch)
        # a read that matches nothing, and a budget of maximal matches that the longest pairs exceed
        q0, l0 = int(seqs["q_off"][3]), int(seqs["q_len"][3])
        seqs["q_arena"] = seqs["q_arena"].copy()
        seqs["q_arena"][q0:q0 + l0] = ord("N")
        sp = api.make_sdp_params(max_fragments=72)
        for kw in (dict(), dict(mode=_lib.MODE_AFFINE, band=16, affine_open=10, affine_extend=1)):
            ctx.set_scoring(api.make_scoring(**kw))
            ctx.set_guide_producer(None)
            status, want, want_blocks = _two_calls(ctx, seqs, sp)
            assert int(status[3]) == _lib.PAIR_EMPTY_GUIDE and (status == _lib.PAIR_TOO_MANY).any() and (status == 0).sum() > 40
            with pytest.raises(api.GbaError):
                ctx.refine_host(seqs)                      # no guides and no producer
            ctx.set_guide_producer(sp)
            got, got_blocks, rc = ctx.refine_host(seqs)
            assert rc == 0
            _same(status, want, want_blocks, got, got_blocks)
            # leftovers in blk_off / n_blk of a batch without blocks are not a guide
            junk = dict(seqs, blk_off=np.full(200, 7, np.uint64), n_blk=np.full(200, 9, np.uint32))
            got2, got_blocks2, rc = ctx.refine_host(junk)
            _same(status, want, want_blocks, got2, got_blocks2)
            # batches that bring guides are not touched by the producer
            again, again_blocks, rc = ctx.refine_host(batch)
            ctx.set_guide_producer(None)
            plain, plain_blocks, rc = ctx.refine_host(batch)
            assert again.tobytes() == plain.tobytes() and np.array_equal(again_blocks, plain_blocks)
        # against the oracles, both stages
        ctx.set_scoring(api.make_scoring())
        ctx.set_guide_producer(sp)
        got, got_blocks, rc = ctx.refine_host(seqs)
        osp = oracle.make_sdp_params(max_fragments=72)
        checked = 0
        for p in range(0, 200, 9):
            q, t, _ = synth.pair(batch, p)
            if p == 3:
                continue
            q, t = bytes(q), bytes(t)
            st, guide, _, _ = oracle.sdp_chain(q, t, osp)
            assert st == int(status[p])
            if st != 0:
                continue
            er, eb = oracle.align(q, t, guide, oracle.make_params())
            r = got[p]
            assert (int(r["status"]), int(r["score"]), int(r["n_blk"]), int(r["q_pos"]), int(r["t_pos"])) == \
                   (0, er["score"], er["n_blk"], er["q_pos"], er["t_pos"])
            kb = got_blocks[int(r["blk_off"]):int(r["blk_off"]) + int(r["n_blk"])]
            assert np.array_equal(kb["q_pos"], eb["q_pos"]) and np.array_equal(kb["t_pos"], eb["t_pos"]) and np.array_equal(kb["length"], eb["length"])
            checked += 1
        assert checked >= 5
    finally:
        ctx.close()


def test_one_call_on_a_batch_that_runs_as_two_halves(api, synth, monkeypatch):
    """A host batch from sequences whose refine stage runs as two halves (400 pairs where a part is 600 and the lane
    classes begin at 100): one call equals two calls, with a block arena and to intervals only, and equals the same
    batch refined in one piece."""
    monkeypatch.setenv("GBA_WAVE_PAIRS", "100")
    monkeypatch.setenv("GBA_PART_PAIRS", "0")
    whole = api.Context(0)
    monkeypatch.setenv("GBA_PART_PAIRS", "600")
    ctx = api.Context(0)
    try:
        batch = synth.generate(400, read_len=1500, seed=1818, len_jitter=900, decoy_every=9)
        seqs = _sequences_only(batch)
        sp = api.make_sdp_params()
        for c in (ctx, whole):
            c.set_scoring(api.make_scoring())
        status, want, want_blocks = _two_calls(ctx, seqs, sp)
        assert (status == 0).sum() > 50 and (status != 0).any()
        for c in (ctx, whole):
            c.set_guide_producer(sp)
            got, got_blocks, rc = c.refine_host(seqs)
            assert rc == 0
            _same(status, want, want_blocks, got, got_blocks)
            only, rc = c.refine_host_intervals(seqs)
            assert rc == 0 and only.tobytes() == got.tobytes()
    finally:
        ctx.close()
        whole.close()


def test_windows_of_the_reference_in_parts_to_intervals(api, synth, monkeypatch):
    """Reads and windows of the resident reference on both strands, a batch run in several parts, no block
    arena: sequences in, interval records out; and the device entry point on the same batch."""
    import torch
    from blasr_amd import _lib
    monkeypatch.setenv("GBA_PART_PAIRS", "1024")
    ctx = api.Context(0)
    try:
        ref = synth.reference(400_000, 5)
        batch = synth.generate(3000, read_len=700, seed=99, len_jitter=300, ref=ref, ref_len=400_000, share_reads=True, decoy_every=4)
        win = synth.as_windows(batch)
        assert win.get("t_arena") is None and win["t_rc"].any()
        ctx.set_reference(ref)
        ctx.set_scoring(api.make_scoring())
        seqs = _sequences_only(win)
        sp = api.make_sdp_params(tuple_size=9)
        status, want, want_blocks = _two_calls(ctx, seqs, sp)
        assert (status == 0).sum() > 700 and (status != 0).sum() > 700      # true windows and decoys
        ctx.set_guide_producer(sp)
        got, got_blocks, rc = ctx.refine_host(seqs)
        assert rc == 0
        _same(status, want, want_blocks, got, got_blocks)
        only, rc = ctx.refine_host_intervals(seqs)
        assert rc == 0 and only.tobytes() == got.tobytes()
        # buffers already on the device
        n = int(seqs["n_pairs"])
        db = api.to_device(seqs)
        assert "blocks" not in db
        out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
        out_blocks = torch.zeros((len(want_blocks) + 8, 3), dtype=torch.int32, device="cuda:0")
        rc, needed = ctx.refine_device(db, out_pairs, out_blocks)
        assert rc == 0 and needed == len(want_blocks)
        dev = np.frombuffer(out_pairs.cpu().numpy().tobytes(), dtype=_lib.PAIR_RESULT_DT)
        dev_blocks = out_blocks.cpu().numpy()[:needed].reshape(-1).view(_lib.BLOCK_DT)
        _same(status, want, want_blocks, dev, dev_blocks)
        with pytest.raises(api.GbaError):
            ctx.refine_device(db, out_pairs, out_blocks, no_alloc=True)
    finally:
        ctx.close()


def test_producer_parameters_are_checked(api):
    ctx = api.Context(0)
    try:
        with pytest.raises(api.GbaError):
            ctx.set_guide_producer(api.make_sdp_params(tuple_size=3))
        with pytest.raises(api.GbaError):
            ctx.set_guide_producer(api.make_sdp_params(max_fragments=5000))
        ctx.set_guide_producer(api.make_sdp_params())
        ctx.set_guide_producer(None)
    finally:
        ctx.close()


def test_host_mirrors_one_call_equals_two_calls(api, synth):
    """refine.AlignAndRefineCandidates against SDPAlignCandidates + RefineAlignments, candidate by candidate;
    the C++ mirror's two forms print the same lines."""
    import subprocess
    from blasr_amd import build, refine
    batch = synth.generate(8, read_len=1800, len_jitter=600, seed=4711)
    ctx = api.Context(0)
    try:
        params = refine.MappingParameters(sortRefinedAlignments=False)
        lines = ["8"]
        for i in range(8):
            q, t, _ = synth.pair(batch, i)
            read, window = bytes(q), bytes(t)
            if i == 5:
                window = b"N" * len(window)             # nothing to anchor on
            lines.append("%s %s 0" % (read.decode(), window.decode()))

            def cands():
                return [refine.AlignmentCandidate(tAlignedSeq=window, tAlignedSeqPos=1000 + i, qAlignedSeqPos=0),
                        refine.AlignmentCandidate(tAlignedSeq=window[40:], tAlignedSeqPos=2000, qAlignedSeqPos=25)]
            two = refine.RefineAlignments(read, refine.SDPAlignCandidates(read, cands(), params, ctx), params, ctx)
            one = refine.AlignAndRefineCandidates(read, cands(), params, ctx)
            iv = refine.AlignAndRefineCandidates(read, cands(), params, ctx, want_blocks=False)
            for a, b, c in zip(two, one, iv):
                assert (a.status == 0) == (b.status == 0) == (c.status == 0) == (i != 5)
                if a.status != 0:
                    assert b.status == c.status == 1 and len(b.blocks) == 0
                    continue
                for f in ("score", "nCells", "nMatch", "nMismatch", "nIns", "nDel", "pctSimilarity", "qPos", "tPos", "qAlignedSeqPos",
                          "tAlignedSeqPos", "qAlignedSeq", "tAlignedSeq", "qAlignedSeqLength", "tAlignedSeqLength", "qInterval", "tInterval", "nBlocks"):
                    assert getattr(a, f) == getattr(b, f) == getattr(c, f), f
                assert np.array_equal(a.blocks, b.blocks) and len(c.blocks) == 0
    finally:
        ctx.close()
    text = "\n".join(lines) + "\n"
    out = [subprocess.run([build.HOST_DEMO, mode], input=text, capture_output=True, text=True, timeout=120) for mode in ("sdp", "fused")]
    assert out[0].returncode == 0 and out[1].returncode == 0, (out[0].stderr, out[1].stderr)
    rows = [o.stdout.strip().split("\n") for o in out]
    assert len(rows[0]) == 8 and rows[0][:5] + rows[0][6:] == rows[1][:5] + rows[1][6:]
    assert rows[0][5].split()[0] != "0" and rows[1][5].split()[0] == "1"      # the pair without anchors


def test_reads_packed_two_bits_per_base(monkeypatch):
    """gba_batch.q_packed: the same batch with its reads as letters and 2 bits per base gives the same bytes back,
    with guides from the caller and with guides made on the device, in one part and in several."""
    from blasr_amd import api, synth
    host = synth.compact_blocks(synth.generate(600, read_len=1500, len_jitter=600, seed=91))
    packed = api.pack_reads_2bit(host)
    assert len(packed["q_arena"]) * 4 >= len(host["q_arena"]) and packed["q_bases"] == len(host["q_arena"])
    for part_pairs in ("0", "97"):
        monkeypatch.setenv("GBA_PART_PAIRS", part_pairs)
        ctx = api.Context(0)
        ctx.set_scoring(api.make_scoring())
        want, wblocks, rc = ctx.refine_host(host)
        assert rc == 0 and int((want["status"] == 0).sum()) == 600
        for b in (packed, ctx.pinned_copy(packed)):
            got, gblocks, rc = ctx.refine_host(b)
            assert rc == 0 and got.tobytes() == want.tobytes() and gblocks.tobytes() == wblocks.tobytes()
        # guides made on the device from the packed reads
        ctx.set_guide_producer(api.make_sdp_params())
        noguide = {k: v for k, v in host.items() if k not in ("blocks", "blk_off", "n_blk")}
        want2, wb2, rc = ctx.refine_host(noguide)
        got2, gb2, rc2 = ctx.refine_host({k: v for k, v in packed.items() if k not in ("blocks", "blk_off", "n_blk")})
        assert rc == 0 and rc2 == 0 and got2.tobytes() == want2.tobytes() and gb2.tobytes() == wb2.tobytes()
        ctx.close()
    # letters that have no 2-bit code are refused by the packer
    bad = dict(host, q_arena=host["q_arena"].copy())
    bad["q_arena"][int(host["q_off"][3]) + 5] = ord("N")
    with pytest.raises(ValueError):
        api.pack_reads_2bit(bad)
