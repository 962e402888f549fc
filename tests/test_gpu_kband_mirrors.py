"""Host mirrors over gba_kband: the -extend step of AlignIntervals, the legacy k-band refine
(PairwiseLocalAlign), and the -global branch with the quality-value scorer.  Each is checked against
the same composition made from the CPU oracle's alignments."""
import numpy as np
import pytest

from kband_cases import _read

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", np.uint8)
COMP = bytes.maketrans(b"ACGT", b"TGCA")


@pytest.fixture(scope="module")
def mods():
    from blasr_amd import api, refine, _lib
    return api, refine, _lib


def _case(rng, strand):
    """A read whose middle 800 bases copy the genome exactly and whose ends are noisy copies of what
    lies on either side; the candidate knows only the middle."""
    genome = rng.choice(ACGT, 6000).tobytes()
    strand_seq = genome if strand == 0 else genome.translate(COMP)[::-1]
    g = np.frombuffer(strand_seq, np.uint8)
    left = _read(rng, g[1000:1600], 0.12, False).tobytes()
    right = _read(rng, g[2400:3000], 0.12, False).tobytes()
    read = left + strand_seq[1600:2400] + right
    return genome, strand_seq, read, len(left)


@pytest.mark.parametrize("strand,t_pos,slack", [(0, 0, 0), (1, 0, 0), (0, 130, 75), (1, 620, 410)])
def test_extend_alignments(mods, oracle, strand, t_pos, slack):
    """t_pos / slack: the candidate's window is wider than its alignment (the normal case after AlignIntervals): t_pos
    bases in front of the first block, slack behind the last; the prelude of BlasrAlignImpl.hpp:1020-1037 cuts both off."""
    api, refine, L = mods
    rng = np.random.default_rng(100 + strand)
    ctx = api.Context(0)
    params = refine.MappingParameters()
    genome, sseq, read, nl = _case(rng, strand)
    assert refine._genome_slice(genome, 1234, 77, strand) == sseq[1234:1234 + 77]
    blk = np.zeros(1, refine.BLOCK_DT); blk[0] = (0, 0, 800)
    c = refine.AlignmentCandidate(tAlignedSeq=sseq[1600 - t_pos:2400 + slack], tAlignedSeqPos=1600 - t_pos, tPos=t_pos, qAlignedSeqPos=0, qPos=nl,
                                  blocks=blk, tStrand=strand, qAlignedSeq=read, qAlignedSeqLength=len(read),
                                  tAlignedSeqLength=800 + t_pos + slack, nMatch=800, score=-4000)
    n = refine.ExtendAlignments(read, [c], genome, params, ctx)
    assert n == 2
    P = refine._kband_params(params, L.KB_EXTEND, ins=params.insertion, dele=params.deletion, max_drops=10)
    fq, ft = read[nl + 800:nl + 800 + 500], sseq[2400:2900]
    fw, fb = oracle.kband_align(fq, ft, 10, P)
    rl = min(nl, 500)
    rq, rt = read[nl - rl:nl], sseq[1600 - 500:1600]
    rv, rb = oracle.kband_align(rq, rt, 10, P, reverse=True)
    assert fw["dp_score"] < 0 and rv["dp_score"] < 0
    # the alignment now starts at the first block of the backward extension and ends with the forward one
    assert c.qAlignedSeqPos == nl - rl and c.tAlignedSeqPos == 1100
    assert (c.qPos, c.tPos) == (rv["q_pos"], rv["t_pos"])
    want = [(int(b["q_pos"]), int(b["t_pos"]), int(b["length"])) for b in rb]
    oq, ot = nl - rl + rv["q_pos"], 1100 + rv["t_pos"]
    want.append((nl - oq, 1600 - ot, 800))
    want += [(nl + 800 + fw["q_pos"] + int(b["q_pos"]) - oq, 2400 + fw["t_pos"] + int(b["t_pos"]) - ot, int(b["length"])) for b in fb]
    got = [(int(b["q_pos"]), int(b["t_pos"]), int(b["length"])) for b in c.blocks]
    assert got == want
    assert c.nMatch == 800 + fw["n_match"] + rv["n_match"] and c.score == -4000 + fw["score"] + rv["score"]
    # every block lies inside the aligned sequences the candidate now holds, and most of its columns agree
    agree = 0
    for q, t, l in got:
        qs = np.frombuffer(read, np.uint8)[c.qAlignedSeqPos + c.qPos + q:][:l]
        ts = np.frombuffer(c.tAlignedSeq, np.uint8)[c.tPos + t:][:l]
        assert len(qs) == l and len(ts) == l
        agree += int((qs == ts).sum())
    assert agree == c.nMatch
    assert c.tAlignedSeq == sseq[c.tAlignedSeqPos:c.tAlignedSeqPos + len(c.tAlignedSeq)] and len(c.tAlignedSeq) == c.tAlignedSeqLength
    assert c.tPos + got[-1][1] + got[-1][2] == len(c.tAlignedSeq) and c.qPos + got[-1][0] + got[-1][2] == c.qAlignedSeqLength
    assert c.qInterval[0] < nl - 100 and c.qInterval[1] > nl + 800 + 100
    ctx.close()


def test_legacy_refine_without_and_with_quality_values(mods, oracle):
    api, refine, L = mods
    rng = np.random.default_rng(7)
    ctx = api.Context(0)
    params = refine.MappingParameters(useGuidedAlign=False, bandSize=0, sortRefinedAlignments=False)
    t = rng.choice(ACGT, 1500)
    read = _read(rng, t[200:1300], 0.12, False).tobytes()
    tw = t.tobytes()
    # sparse "SDP" blocks: exact 12-mers shared by read and window, picked along the true diagonal drift
    import difflib
    sm = difflib.SequenceMatcher(None, read, tw, autojunk=False)
    mb = [m for m in sm.get_matching_blocks() if m.size >= 12]
    blk = np.zeros(len(mb), refine.BLOCK_DT)
    for i, m in enumerate(mb):
        blk[i] = (m.a - mb[0].a, m.b - mb[0].b, m.size)
    mk = lambda: refine.AlignmentCandidate(tAlignedSeq=tw, qAlignedSeqPos=0, qPos=mb[0].a, tPos=mb[0].b, blocks=blk.copy())
    c = mk()
    refine.RefineAlignments(read, [c], params, ctx)
    last = blk[-1]
    ql, tl = int(last["q_pos"]) + int(last["length"]), int(last["t_pos"]) + int(last["length"])
    k = max(int(refine.compute_drift(blk) * 1.5), 10)
    qs, ts = read[mb[0].a:mb[0].a + ql], tw[mb[0].b:mb[0].b + tl]
    want, wb = oracle.kband_align(qs, ts, int(k * 1.2), refine._kband_params(params, L.KB_GLOBAL, L.KB_AFFINE_HP, ins=5, dele=5))
    assert c.status == 0 and want["status"] == 0
    assert (c.score, c.nMatch, c.nIns, c.nDel, len(c.blocks)) == (want["score"], want["n_match"], want["n_ins"], want["n_del"], want["n_blk"])
    assert (c.blocks == wb).all() and c.qAlignedSeqPos == mb[0].a and c.tAlignedSeqPos == mb[0].b
    # the same candidate with quality values: KBandAlign with the IDS scorer, Fit
    n = len(read)
    qv = refine.QualityValues(qual=bytes(rng.integers(1, 40, n, dtype=np.uint8)), insertionQV=bytes(rng.integers(1, 20, n, dtype=np.uint8)),
                              deletionQV=bytes(rng.integers(1, 20, n, dtype=np.uint8)), deletionTag=rng.choice(np.frombuffer(b"ACGTN", np.uint8), n).tobytes(),
                              substitutionQV=bytes(rng.integers(1, 30, n, dtype=np.uint8)), substitutionTag=rng.choice(ACGT, n).tobytes())
    c2 = mk()
    refine.refine_legacy(read, [c2], params, ctx, qv=qv)
    cut = lambda b: b[mb[0].a:mb[0].a + ql]
    prof = refine.ids_profile(qs, refine.QualityValues(*[cut(getattr(qv, f)) for f in
                                                         ("qual", "insertionQV", "deletionQV", "deletionTag", "substitutionQV", "substitutionTag")]), params)
    want, wb = oracle.kband_align(qs, ts, k, refine._kband_params(params, L.KB_FIT, ins=7, dele=7), profile=prof)
    assert c2.status == 0 and (c2.score, c2.nMatch, len(c2.blocks), c2.tPos) == (want["score"], want["n_match"], want["n_blk"], want["t_pos"])
    assert (c2.blocks == wb).all()
    ctx.close()


def test_global_branch_with_quality_values(mods, oracle):
    api, refine, L = mods
    rng = np.random.default_rng(19)
    ctx = api.Context(0)
    params = refine.MappingParameters(doGlobalAlignment=True)
    t = rng.choice(ACGT, 900)
    read = _read(rng, t, 0.1, False).tobytes()
    n = len(read)
    qv = refine.QualityValues(qual=bytes(rng.integers(1, 40, n, dtype=np.uint8)), insertionQV=bytes(rng.integers(1, 20, n, dtype=np.uint8)),
                              deletionQV=bytes(rng.integers(1, 20, n, dtype=np.uint8)), deletionTag=rng.choice(np.frombuffer(b"ACGTN", np.uint8), n).tobytes(),
                              substitutionQV=bytes(rng.integers(1, 30, n, dtype=np.uint8)), substitutionTag=rng.choice(ACGT, n).tobytes())
    blk = np.zeros(2, refine.BLOCK_DT); blk[0] = (0, 0, 12); blk[1] = (400, 400 + len(t) - n + 40, 12)
    c = refine.AlignmentCandidate(tAlignedSeq=t.tobytes(), blocks=blk)
    k = refine.compute_drift(blk)
    refine.refine_global_qv(read, [c], params, ctx, qv)
    want, wb = oracle.kband_align(read, t.tobytes(), k, refine._kband_params(params, L.KB_GLOBAL), profile=refine.ids_profile(read, qv, params))
    assert want["status"] == 0 and c.status == 0
    assert (c.score, c.qPos, c.tPos, len(c.blocks)) == (want["score"], want["q_pos"], want["t_pos"], want["n_blk"])
    assert (c.blocks == wb).all()
    ctx.close()


def test_cpp_shim_legacy_refine_and_extend(mods, oracle):
    """refine_demo legacy / extend: gba::RefineLegacy and gba::ExtendAlignments of gba_refine.hpp give what
    the Python mirrors give on the same candidates."""
    import os
    import subprocess
    from blasr_amd import build
    api, refine, L = mods
    exe = build.HOST_DEMO
    assert os.path.exists(exe), "run python -m blasr_amd.build"
    rng = np.random.default_rng(77)
    ctx = api.Context(0)
    # -- legacy: four reads with sparse exact blocks as guides
    import difflib
    params = refine.MappingParameters(useGuidedAlign=False, bandSize=0, sortRefinedAlignments=False)
    lines, want = ["4"], []
    for _ in range(4):
        t = rng.choice(ACGT, 900)
        read = _read(rng, t[100:800], 0.1, False).tobytes()
        tw = t.tobytes()
        mb = [m for m in difflib.SequenceMatcher(None, read, tw, autojunk=False).get_matching_blocks() if m.size >= 12]
        blk = np.zeros(len(mb), refine.BLOCK_DT)
        for i, m in enumerate(mb):
            blk[i] = (m.a - mb[0].a, m.b - mb[0].b, m.size)
        # the demo's candidates start at (0, 0) of both strings: hand it the strings cut at the first block
        rd, tg = read[mb[0].a:], tw[mb[0].b:]
        c = refine.AlignmentCandidate(tAlignedSeq=tg, blocks=blk.copy())
        refine.refine_legacy(rd, [c], params, ctx)
        want.append(c)
        lines.append("%s %s %d %s" % (rd.decode(), tg.decode(), len(blk), " ".join("%d %d %d" % tuple(int(v) for v in b) for b in blk)))
    out = subprocess.run([exe, "legacy"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    for row, c in zip(out.stdout.strip().split("\n"), want):
        v = [int(x) for x in row.split()]
        assert v[:6] == [c.status, c.score, c.nCells, c.qPos, c.tPos, len(c.blocks)]
        assert v[6:] == [int(x) for b in c.blocks for x in b]
    # -- extend: the candidate of test_extend_alignments on the forward strand, its window cut to the alignment; the same
    # with slack on both sides of the alignment inside a wider window (tPos = 150, then tPos = 700 > the 500 bases an
    # extension may take), and the legacy branch's interval fields
    genome, sseq, read, nl = _case(rng, 0)
    lines, want = [], []
    for t_pos, slack in ((0, 0), (150, 90), (700, 300)):
        blk = np.zeros(1, refine.BLOCK_DT); blk[0] = (0, 0, 800)
        w0 = 1600 - t_pos
        c = refine.AlignmentCandidate(tAlignedSeq=sseq[w0:2400 + slack], tAlignedSeqPos=w0, qAlignedSeqPos=0, qPos=nl, tPos=t_pos, blocks=blk,
                                      qAlignedSeq=read, qAlignedSeqLength=len(read), tAlignedSeqLength=2400 + slack - w0)
        assert refine.ExtendAlignments(read, [c], genome, refine.MappingParameters(), ctx) == 2
        want.append(c)
        lines.append("%s %d %d 0 %d %d 1 0 0 800" % (read.decode(), w0, 2400 + slack - w0, nl, t_pos))
    for c in want[1:]:          # slack in the window changes nothing: the prelude cuts it off
        assert (c.qAlignedSeqPos, c.tAlignedSeqPos, c.qPos, c.tPos, c.tAlignedSeq) == (want[0].qAlignedSeqPos, want[0].tAlignedSeqPos, want[0].qPos, want[0].tPos, want[0].tAlignedSeq)
        assert (c.blocks == want[0].blocks).all()
    text = "%d\n%s\n%s\n" % (len(lines), genome.decode(), "\n".join(lines))
    out = subprocess.run([exe, "extend"], input=text, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    for row, c in zip(out.stdout.strip().split("\n"), want):
        f = row.split()
        v = [int(x) for x in f[:10]]
        assert v == [0, c.score, c.qAlignedSeqPos, c.tAlignedSeqPos, c.qPos, c.tPos, c.qInterval[0], c.qInterval[1], c.tInterval[0], c.tInterval[1]]
        assert f[10] == c.tAlignedSeq.decode()
        assert [int(x) for x in f[11:]] == [len(c.blocks)] + [int(x) for b in c.blocks for x in b]
        # the aligned target string is the genome between the alignment's boundaries, every block inside it
        assert c.tAlignedSeq == sseq[c.tAlignedSeqPos:c.tAlignedSeqPos + len(c.tAlignedSeq)]
        last = c.blocks[-1]
        assert c.tPos + int(last["t_pos"]) + int(last["length"]) == len(c.tAlignedSeq)
        assert c.qPos + int(last["q_pos"]) + int(last["length"]) == c.qAlignedSeqLength == len(c.qAlignedSeq)
    ctx.close()
