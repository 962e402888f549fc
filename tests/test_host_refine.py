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
    a = refine.scoring_from_params(refine.MappingParameters(affineAlign=True))
    assert (a.band, a.mode, a.stats_affine) == (16, _lib.MODE_AFFINE, 1)


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
