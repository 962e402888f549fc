"""The per-read filters MapRead runs behind the refine stage (/root/reference/iblasr/BlasrAlignImpl.hpp:381-382):
RemoveLowQualityAlignments (iblasr/BlasrUtilsImpl.hpp:476-519) and RemoveOverlappingAlignments (:523-605), the
Python mirror (blasr_amd/refine.py) on hand cases and against the C++ mirror (gba_refine.hpp) on random ones."""
import subprocess

import numpy as np

from blasr_amd import refine
from blasr_amd.synth import BLOCK_DT


def _cand(t0, t1, score, t_index=0, pct=90.0, blocks=True):
    b = np.zeros(1 if blocks else 0, dtype=BLOCK_DT)
    if blocks:
        b[0] = (0, 0, t1 - t0)
    return refine.AlignmentCandidate(tAlignedSeq=b"", tAlignedSeqPos=t0, blocks=b, score=score, tIndex=t_index, pctSimilarity=pct)


def test_contained_alignment_goes_when_its_score_is_no_better():
    P = refine.MappingParameters()
    outer, inner_worse, inner_better, other_contig, apart = (_cand(100, 1100, -900), _cand(200, 900, -500), _cand(300, 800, -950),
                                                             _cand(200, 900, -100, t_index=1), _cand(5000, 6000, -300))
    cands = [outer, inner_worse, inner_better, other_contig, apart]
    assert refine.RemoveOverlappingAlignments(cands, P) == 4
    same = lambda xs, ys: [id(x) for x in xs] == [id(y) for y in ys]
    assert same(cands, [outer, inner_better, other_contig, apart])       # order kept; the better inner one stays (:565)
    # the inner one comes first in the list and the outer one scores no worse: the inner one goes (:575-586)
    a, b = _cand(200, 900, -800), _cand(100, 1100, -800)
    cands = [a, b]
    refine.RemoveOverlappingAlignments(cands, P)
    assert same(cands, [b])
    # ... and stays when it scores better than the one that holds it
    a, b = _cand(200, 900, -900), _cand(100, 1100, -800)
    cands = [a, b]
    refine.RemoveOverlappingAlignments(cands, P)
    assert same(cands, [a, b])
    # identical stretches and scores: the first branch (:559) holds, the later one goes
    a, b = _cand(100, 1100, -800), _cand(100, 1100, -800)
    cands = [a, b]
    refine.RemoveOverlappingAlignments(cands, P)
    assert same(cands, [a])
    # a candidate below minPctSimilarity is never the one others are measured against (:538), but can still be dropped
    low, inner = _cand(100, 1100, -900, pct=40.0), _cand(200, 900, -100)
    cands = [low, inner]
    refine.RemoveOverlappingAlignments(cands, refine.MappingParameters(minPctSimilarity=70.0))
    assert same(cands, [low, inner])
    assert refine.RemoveOverlappingAlignments([], P) == 0


def test_low_quality_filter_looks_at_the_first_ncandidates_only():
    P = refine.MappingParameters(maxScore=-200, nCandidates=3)
    cands = [_cand(0, 10, -900), _cand(0, 10, -500), _cand(0, 10, -300), _cand(0, 10, -100), _cand(0, 10, -50)]
    assert len(refine.RemoveLowQualityAlignments(list(cands), P)) == 5          # the bad ones lie beyond the first three
    P.nCandidates = 10
    assert len(refine.RemoveLowQualityAlignments(list(cands), P)) == 3
    cands[1] = _cand(0, 10, -500, blocks=False)
    assert len(refine.RemoveLowQualityAlignments(list(cands), P)) == 1          # an alignment without blocks ends the list


def test_cpp_filters_equal_python():
    from blasr_amd import build
    exe = build.build_host_demo()
    rng = np.random.default_rng(23)
    for trial in range(40):
        n = int(rng.integers(0, 12))
        min_pct = float(rng.choice([0.0, 75.0]))
        ncand = int(rng.choice([10, 4]))
        cands, lines = [], ["%d" % n]
        for _ in range(n):
            t0 = int(rng.integers(0, 40)) * 100
            t1 = t0 + int(rng.integers(1, 30)) * 100
            score = -int(rng.integers(1, 8)) * 100
            ti = int(rng.integers(0, 2))
            pct = float(rng.choice([60.0, 85.0]))
            nb = int(rng.random() < 0.9)
            cands.append(_cand(t0, t1, score, ti, pct, blocks=bool(nb)))
            lines.append("%d %d %d %d %g %d" % (ti, t0, t1, score, pct, nb))
        cands.sort(key=lambda c: c.score)
        order = {id(c): i for i, c in enumerate(cands)}
        lines = [lines[0]] + ["%d %d %d %d %g %d" % (c.tIndex, c.tAlignedSeqPos, c.tAlignedSeqPos + (int(c.blocks[0]["length"]) if len(c.blocks) else 0),
                                                       c.score, c.pctSimilarity, len(c.blocks)) for c in cands]
        P = refine.MappingParameters(minPctSimilarity=min_pct, nCandidates=ncand)
        left = list(cands)
        refine.RemoveLowQualityAlignments(left, P)
        refine.RemoveOverlappingAlignments(left, P)
        out = subprocess.run([exe, "filters", "%g" % min_pct, "%d" % ncand], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        assert [int(x) for x in out.stdout.split()] == [order[id(c)] for c in left], trial
