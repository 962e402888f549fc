"""Host logic of the refine loop (blasr_amd/refine.py, blasr_amd/host/gba_refine.hpp)."""
import os
import subprocess

import numpy as np
import pytest

from blasr_amd import _lib, emit, refine
from blasr_amd.synth import BLOCK_DT


def _blocks(lst):
    a = np.zeros(len(lst), dtype=BLOCK_DT)
    for i, b in enumerate(lst):
        a[i] = b
    return a


def test_scoring_from_params_defaults():
    s = refine.scoring_from_params(refine.MappingParameters())
    # distScoreFn(M, deletion=5, insertion=4) read as ctor(ins, del): BlasrUtilsImpl.hpp:630-631
    assert (s.ins, s.dele, s.stats_ins, s.stats_del) == (5, 4, 5, 5)
    assert (s.band, s.mode) == (10, _lib.MODE_LINEAR)
    assert s.sub[0][0] == -5 and s.sub[0][1] == 6 and s.sub[4][2] == 0
    a = refine.scoring_from_params(refine.MappingParameters(affineAlign=True, affineOpen=7, affineExtend=3))
    assert (a.band, a.mode, a.stats_affine) == (16, _lib.MODE_AFFINE, 1)
    # RefineAlignment never hands params.affineOpen / affineExtend to the distance-matrix scorers
    # (BlasrUtilsImpl.hpp:630-634 against :643-644): constructor defaults, named in refine.py
    assert (a.affine_open, a.affine_extend) == (refine.DIST_SCORER_AFFINE_OPEN, refine.DIST_SCORER_AFFINE_EXTEND) == (0, 0)
    assert (a.stats_affine_open, a.stats_affine_extend) == (0, 0)
    with pytest.raises(ValueError):
        refine.scoring_from_params(refine.MappingParameters(doGlobalAlignment=True))
    g = refine.scoring_global(refine.MappingParameters(), 37)
    # KBandAlign(..., params.insertion, params.deletion, drift, ...) BlasrUtilsImpl.hpp:657-661
    assert (g.mode, g.kband_k, g.ins, g.dele, g.stats_ins, g.stats_affine) == (_lib.MODE_KBAND, 37, 4, 5, 5, 0)


def test_compute_drift_and_remove_end_gaps():
    assert refine.compute_drift(_blocks([])) == 0
    assert refine.compute_drift(_blocks([(0, 0, 10), (12, 15, 5), (30, 28, 4)])) == 3
    assert refine.compute_drift(_blocks([(5, 9, 10), (20, 20, 5)])) == 4
    c = refine.AlignmentCandidate(tAlignedSeq=b"", blocks=_blocks([(2, 5, 4), (8, 11, 3)]), qPos=1, tPos=10)
    refine.remove_end_gaps(c)
    assert (c.qPos, c.tPos) == (3, 15) and [tuple(b) for b in c.blocks] == [(0, 0, 4), (6, 6, 3)]


def test_build_batch_slicing_rules():
    read = b"NNNNN" + b"ACGTACGTACGTACGTACGT" + b"NN"
    win = b"GG" + b"ACGTACGTACGTTTACGTACGT" + b"CCCC"
    c = refine.AlignmentCandidate(tAlignedSeq=win, qAlignedSeqPos=3, qPos=2, tPos=2,
                                  blocks=_blocks([(0, 0, 12), (12, 14, 8)]))
    empty = refine.AlignmentCandidate(tAlignedSeq=win)
    b = refine.build_batch(read, [c, empty])
    # query: VIEW read[qAlignedSeqPos+qPos : +lastBlock.qPos+len]  (:700-703)
    assert (int(b["q_off"][0]), int(b["q_len"][0])) == (5, 20)
    # target: COPY tAlignedSeq[tPos : tPos + lastBlock.tEnd - blocks[0].tPos]  (:693-697)
    assert bytes(b["t_arena"][int(b["t_off"][0]):int(b["t_off"][0]) + int(b["t_len"][0])]) == win[2:2 + 22]
    assert int(b["n_blk"][1]) == 0


def test_copy_back_and_filters():
    read = b"ACGTACGTACGTACGTACGT"
    c = refine.AlignmentCandidate(tAlignedSeq=b"TT" + read, tAlignedSeqPos=100, qAlignedSeqPos=0, tPos=2,
                                  blocks=_blocks([(0, 0, 20)]))
    batch = refine.build_batch(read, [c])
    pairs = np.zeros(1, dtype=_lib.PAIR_RESULT_DT)
    pairs[0]["n_blk"], pairs[0]["score"], pairs[0]["ncells"] = 1, -300, 440
    pairs[0]["n_match"], pairs[0]["rows"], pairs[0]["cols"] = 20, 20, 20
    refine.copy_back(read, [c], batch, pairs, _blocks([(0, 0, 20)]))
    assert (c.score, c.nCells, c.tAlignedSeqPos, c.tPos, c.tAlignedSeqLength) == (-300, 440, 102, 0, 20)
    assert c.pctSimilarity == 100.0 and c.tAlignedSeq == read
    p = refine.MappingParameters()
    weak = refine.AlignmentCandidate(tAlignedSeq=b"", score=-10, blocks=_blocks([(0, 0, 1)]))
    assert refine.RemoveLowQualityAlignments([c, weak, c], p) == [c]     # cut at the first bad one


def _oracle_expect(oracle, q, t, blk):
    res, blocks = oracle.align(q, t, blk, oracle.make_params())
    return res, blocks


@pytest.mark.gpu
def test_refine_alignments_end_to_end(oracle, synth):
    from blasr_amd import api
    ctx = api.Context(0)
    batch = synth.generate(12, read_len=900, seed=123)
    p = refine.MappingParameters()
    q0, _, _ = synth.pair(batch, 0)
    read = b"GATTACA" + bytes(q0) + b"TTTT"          # the candidate's query sits inside a longer read
    cands, expect = [], []
    for i in range(12):
        q, t, blk = synth.pair(batch, i)
        # every candidate views read 0 (shared query); use its own window with a two-block guide
        L = min(len(q0), len(t))
        g = _blocks([(0, 0, 11), (L - 11, L - 11, 11)])
        cands.append(refine.AlignmentCandidate(tAlignedSeq=b"AC" + bytes(t) + b"GG", tAlignedSeqPos=50 + i,
                                               qAlignedSeqPos=7, tPos=2, blocks=g))
        expect.append(oracle.align(bytes(q0)[:L], bytes(t)[:L], g, oracle.make_params(band=10)))
    cands.append(refine.AlignmentCandidate(tAlignedSeq=b"ACGT"))           # no blocks: untouched
    out = refine.RefineAlignments(read, list(cands), p, ctx)
    assert [c.score for c in out] == sorted(c.score for c in out)           # re-sorted by score
    for c, (er, eb) in zip(cands[:12], expect):
        assert c.status == 0 and (c.score, c.nCells, c.nMatch) == (er["score"], er["ncells"], er["n_match"])
        assert np.array_equal(c.blocks["length"], eb["length"]) and np.array_equal(c.blocks["t_pos"], eb["t_pos"])
        assert c.tAlignedSeqPos >= 52 and c.qAlignedSeqPos == 7
        # the text record of the refined candidate agrees with the statistics the GPU computed
        ops = emit.cigar_ops(c.blocks, c.qAlignedSeq, c.tAlignedSeq, q_pos=c.qPos, t_pos=c.tPos, allow_adjacent_indels=True)
        assert sum(n for n, op in ops if op == "=") == c.nMatch and sum(n for n, op in ops if op == "X") == c.nMismatch
        assert len(emit.m4_line(c, "read", "ref", len(read), 10000).split()) == 13
    assert cands[12].status == _lib.PAIR_EMPTY_GUIDE and len(cands[12].blocks) == 0
    ctx.close()


def _global_expect(oracle, sub, t, blk, p):
    """What the -global branch must leave in a candidate, from the oracle's k-band alignment."""
    k = refine.compute_drift(blk)
    P = oracle.make_params(mode=oracle.MODE_KBAND, kband_k=k, ins=p.insertion, dele=p.deletion, indel=p.indel)
    r, b = oracle.align(sub, t, [], P)
    if r["status"] != 0:
        return r["status"], None
    dq, dt = (int(b[0]["q_pos"]), int(b[0]["t_pos"])) if len(b) else (0, 0)
    return 0, (r["score"], dq, dt, [(int(x["q_pos"]) - dq, int(x["t_pos"]) - dt, int(x["length"])) for x in b])


@pytest.mark.gpu
def test_refine_global_branch(oracle, synth):
    """-global (BlasrUtilsImpl.hpp:648-675): whole subread against the whole window, k = ComputeDrift,
    RemoveEndGaps, qPos += SubreadStart; Python mirror and C++ shim against the oracle."""
    from blasr_amd import api, build
    ctx = api.Context(0)
    batch = synth.generate(10, read_len=600, len_jitter=200, seed=777)
    p = refine.MappingParameters(doGlobalAlignment=True, sortRefinedAlignments=False)
    lines = ["10"]
    for i in range(10):
        q, t, blk = synth.pair(batch, i)
        q, t = bytes(q), bytes(t)
        if i % 3 == 1:
            t = b"ACGTTGCA" + t            # a leading target-only stretch: the alignment starts with a gap
            blk = blk.copy(); blk["t_pos"] += 8
        read = b"TTTTT" + q + b"GG"
        c = refine.AlignmentCandidate(tAlignedSeq=t, blocks=blk.copy(), score=12345, nCells=7, tAlignedSeqPos=99)
        refine.refine_global(read, [c], p, ctx, subread_start=5, subread_length=len(q))
        st, e = _global_expect(oracle, q, t, blk, p)
        assert c.status == st
        lines.append("%s %s %d %s" % (q.decode(), t.decode(), len(blk), " ".join("%d %d %d" % tuple(b) for b in blk)))
        if st != 0:
            assert c.score == 12345
            continue
        score, dq, dt, eb = e
        assert (c.score, c.qPos, c.tPos) == (score, dq + 5, dt)
        assert [tuple(int(v) for v in b) for b in c.blocks] == eb
        assert (c.nCells, c.tAlignedSeqPos, c.tAlignedSeq) == (7, 99, t)       # untouched by this branch
    ctx.close()
    out = subprocess.run([build.HOST_DEMO, "global"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    for i, row in enumerate(out.stdout.strip().split("\n")):
        v = [int(x) for x in row.split()]
        q, t, nb = lines[i + 1].split()[:3]
        blk = _blocks([tuple(int(x) for x in lines[i + 1].split()[3 + 3 * j:6 + 3 * j]) for j in range(int(nb))])
        st, e = _global_expect(oracle, q.encode(), t.encode(), blk, p)
        assert v[0] == st
        if st == 0:
            score, dq, dt, eb = e
            assert (v[1], v[3], v[4], v[5]) == (score, dq, dt, len(eb))
            assert [tuple(x) for x in np.array(v[6:]).reshape(-1, 3)] == eb


@pytest.mark.gpu
def test_cpp_shim_with_one_reused_read_buffer(oracle, synth):
    """A caller that hands every read to RefineBatch::Add through one std::string it overwrites
    (equal lengths: same data pointer, same size) must still get every read aligned as itself."""
    from blasr_amd import build
    batch = synth.generate(6, read_len=400, seed=55)
    L = int(batch["q_len"].min())
    lines = ["6"]
    cases = []
    for i in range(6):
        q, t, blk = synth.pair(batch, i)
        q = bytes(q)[:L]
        blk = np.array([b for b in blk if int(b["q_pos"]) + int(b["length"]) <= L], dtype=BLOCK_DT)
        tl = int(blk[-1]["t_pos"]) + int(blk[-1]["length"])
        cases.append((q, bytes(t)[:tl], blk))
        lines.append("%s %s %d %s" % (q.decode(), bytes(t).decode(), len(blk), " ".join("%d %d %d" % tuple(b) for b in blk)))
    out = subprocess.run([build.HOST_DEMO, "reuse"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    for (q, t, blk), row in zip(cases, out.stdout.strip().split("\n")):
        v = [int(x) for x in row.split()]
        er, eb = oracle.align(q, t, blk, oracle.make_params())
        assert v[:6] == [0, er["score"], er["ncells"], er["q_pos"], er["t_pos"], er["n_blk"]]


@pytest.mark.gpu
def test_cpp_host_shim_matches_oracle(oracle, synth):
    from blasr_amd import build
    exe = build.HOST_DEMO
    assert os.path.exists(exe), "run python -m blasr_amd.build"
    batch = synth.generate(20, read_len=700, len_jitter=300, seed=321)
    lines = ["20"]
    for i in range(20):
        q, t, blk = synth.pair(batch, i)
        lines.append("%s %s %d %s" % (bytes(q).decode(), bytes(t).decode(), len(blk),
                                      " ".join("%d %d %d" % (b["q_pos"], b["t_pos"], b["length"]) for b in blk)))
    out = subprocess.run([exe, "cigar"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = out.stdout.strip().split("\n")
    assert len(rows) == 20
    for i, row in enumerate(rows):
        *nums, cig_merged, cig_adjacent = row.split()
        v = [int(x) for x in nums]
        q, t, blk = synth.pair(batch, i)
        er, eb = oracle.align(q, t, blk, oracle.make_params())
        assert v[:6] == [0, er["score"], er["ncells"], er["q_pos"], er["t_pos"], er["n_blk"]]
        got = np.array(v[6:]).reshape(-1, 3)
        assert np.array_equal(got[:, 0], eb["q_pos"]) and np.array_equal(got[:, 2], eb["length"])
        # the C++ shim's CIGAR is the Python one's, in both forms
        kw = dict(q_pos=int(er["q_pos"]), t_pos=int(er["t_pos"]))
        assert cig_merged == emit.cigar(eb, bytes(q), bytes(t), **kw)
        assert cig_adjacent == emit.cigar(eb, bytes(q), bytes(t), allow_adjacent_indels=True, **kw)


@pytest.mark.gpu
def test_cpp_guides_then_refine_matches_the_oracles(oracle, synth):
    """refine_demo sdp: gba::SDPAlignCandidates makes the guide of every candidate, RefineBatch refines it;
    against the two oracles run one after the other on the same sequences."""
    from blasr_amd import build
    exe = build.HOST_DEMO
    assert os.path.exists(exe), "run python -m blasr_amd.build"
    batch = synth.generate(8, read_len=1500, len_jitter=500, seed=77)
    lines = ["8"]
    for i in range(8):
        q, t, _ = synth.pair(batch, i)
        lines.append("%s %s 0" % (bytes(q).decode(), bytes(t).decode()))
    out = subprocess.run([exe, "sdp"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = out.stdout.strip().split("\n")
    assert len(rows) == 8
    for i, row in enumerate(rows):
        v = [int(x) for x in row.split()]
        q, t, _ = synth.pair(batch, i)
        q, t = bytes(q), bytes(t)
        st, guide, _, _ = oracle.sdp_chain(q, t, oracle.make_sdp_params(max_fragments=1024))
        assert st == 0
        q0, t0 = int(guide[0]["q_pos"]), int(guide[0]["t_pos"])
        g = guide.copy(); g["q_pos"] -= np.uint32(q0); g["t_pos"] -= np.uint32(t0)
        ql, tl = int(g[-1]["q_pos"]) + int(g[-1]["length"]), int(g[-1]["t_pos"]) + int(g[-1]["length"])
        er, eb = oracle.align(q[q0:q0 + ql], t[t0:t0 + tl], g, oracle.make_params())
        assert v[:6] == [0, er["score"], er["ncells"], er["q_pos"], er["t_pos"], er["n_blk"]]
        got = np.array(v[6:]).reshape(-1, 3)
        assert np.array_equal(got[:, 0], eb["q_pos"]) and np.array_equal(got[:, 1], eb["t_pos"]) and np.array_equal(got[:, 2], eb["length"])


def test_dense_anchors_bypass_the_guide_producer():
    """AlignIntervals' shortcut: anchors that cover enough of the read are the guide (no device call)."""
    from blasr_amd import refine
    read = b"ACGT" * 50
    c = refine.AlignmentCandidate(tAlignedSeq=b"ACGT" * 60, tAlignedSeqPos=1000, qAlignedSeqPos=10)
    anchors = [[(10, 1000, 80), (100, 1095, 90)]]
    params = refine.MappingParameters(sdpBypassThreshold=0.75)
    refine.GuidesFromAnchorsOrSDP(read, [c], anchors, params, ctx=None)
    assert [tuple(int(v) for v in b) for b in c.blocks] == [(0, 0, 80), (90, 95, 90)] and c.status == 0


@pytest.mark.gpu
def test_cpp_worker_pool_equals_one_context(synth):
    """refine_demo pool: gba::HostPipeline (two contexts on the GPU, two batches in flight) and gba::MultiGpuRefiner (three
    workers of two contexts each, contiguous ranges of whole reads, block arenas guessed too small at first) return the
    bytes one context returns, and every worker stages only its share of the arenas."""
    from blasr_amd import build
    exe = build.HOST_DEMO
    assert os.path.exists(exe), "run python -m blasr_amd.build"
    batch = synth.generate(96, read_len=900, len_jitter=400, seed=99)
    lines = ["96"]
    for i in range(96):
        q, t, blk = synth.pair(batch, i)
        lines.append("%s %s %d %s" % (bytes(q).decode(), bytes(t).decode(), len(blk),
                                      " ".join("%d %d %d" % (b["q_pos"], b["t_pos"], b["length"]) for b in blk)))
    out = subprocess.run([exe, "pool"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    f = out.stdout.split()
    assert f[0] == "ok" and int(f[1]) == 96 and int(f[2]) > 96
    staged, total = [int(x) for x in f[3:6]], int(f[6])
    assert sum(staged) == total and max(staged) < 0.5 * total


@pytest.mark.gpu
def test_guides_between_anchors(oracle, synth):
    """The -rbao guide (refine.GuidesBetweenAnchors, gba::GuidesBetweenAnchors: /root/reference/iblasr/BlasrAlignImpl.hpp:785-978):
    anchors plus the chain found inside every gap between two of them, against the same composition made from the SDP
    oracle's chains; an anchor without a two-sided gap behind it is left out as the reference leaves it out; the guide then
    drives the refine stage; the C++ mirror gives the same candidates."""
    from blasr_amd import api, build, refine
    import difflib
    batch = synth.generate(5, read_len=2500, len_jitter=500, seed=404)
    ctx = api.Context(0)
    params = refine.MappingParameters(sortRefinedAlignments=False, refineBetweenAnchorsOnly=True)
    try:
        lines, got = ["5"], []
        for p in range(5):
            q, t, _ = synth.pair(batch, p)
            read, window = bytes(q), bytes(t)
            # anchors: long exact matches in order, as seeding would leave them (every third one kept: wide gaps between them)
            mb = [m for m in difflib.SequenceMatcher(None, read, window, autojunk=False).get_matching_blocks() if m.size >= 14][::3]
            anchors = [(m.a, m.b, m.size) for m in mb]
            if p == 1 and len(anchors) > 3:           # an anchor that touches the next one in the target: no gap on that side
                a2 = anchors[2]
                anchors[1] = (anchors[1][0], anchors[1][1], a2[1] - anchors[1][1]) if a2[1] - anchors[1][1] <= a2[0] - anchors[1][0] else anchors[1]
            c = refine.AlignmentCandidate(tAlignedSeq=window)
            refine.GuidesFromAnchorsOrSDP(read, [c], [anchors], params, ctx)
            # the same from the oracle's chains
            want = []
            for k in range(len(anchors) - 1):
                qa, ta, la = anchors[k]
                q_gap, t_gap = anchors[k + 1][0] - (qa + la), anchors[k + 1][1] - (ta + la)
                if q_gap > 0 and t_gap > 0:
                    want.append((qa, ta, la))
                    st, g, _, _ = oracle.sdp_chain(read[qa + la:qa + la + q_gap], window[ta + la:ta + la + t_gap], oracle.make_sdp_params())
                    if st == 0:
                        want += [(int(b["q_pos"]) + qa + la, int(b["t_pos"]) + ta + la, int(b["length"])) for b in g]
            want.append(anchors[-1])
            assert c.status == 0 and (c.qPos, c.tPos) == (want[0][0], want[0][1])
            assert [(int(b["q_pos"]) + c.qPos, int(b["t_pos"]) + c.tPos, int(b["length"])) for b in c.blocks] == want, p
            assert len(want) > len([1 for k in range(len(anchors))])       # the gaps did contribute blocks
            guide = c.blocks.copy()
            refine.RefineAlignments(read, [c], params, ctx)
            ql, tl = int(guide[-1]["q_pos"] + guide[-1]["length"]), int(guide[-1]["t_pos"] + guide[-1]["length"])
            er, eb = oracle.align(read[want[0][0]:want[0][0] + ql], window[want[0][1]:want[0][1] + tl], guide, oracle.make_params())
            assert c.status == 0 and (c.score, c.nCells, len(c.blocks)) == (er["score"], er["ncells"], er["n_blk"])
            got.append(c)
            lines.append("%s %s 0 %d %s" % (read.decode(), window.decode(), len(anchors), " ".join("%d %d %d" % a for a in anchors)))
        out = subprocess.run([build.HOST_DEMO, "rbao"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        for row, c in zip(out.stdout.strip().split("\n"), got):
            v = [int(x) for x in row.split()]
            assert v[:6] == [c.status, c.score, c.nCells, c.qPos, c.tPos, len(c.blocks)]
            assert v[6:] == [int(x) for b in c.blocks for x in b]
    finally:
        ctx.close()
