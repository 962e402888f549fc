"""Host-side mirror of the reference's refine loop over the C-ABI.

Mirrors, name for name, what `RefineAlignments` / `RefineAlignment` do for the default
(guided) branch and the `-global` branch (/root/reference/iblasr/BlasrUtilsImpl.hpp:620-923):
slice the sequences to the span of the guide, pick scorer and band from MappingParameters,
align (here: one batched call into libgba instead of one GuidedAlign call per candidate),
copy blocks / score / nCells / positions back, re-sort by score.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from .api import make_scoring, make_sdp_params

BLOCK_DT = _lib.BLOCK_DT


@dataclass
class MappingParameters:
    """Hot-path subset of iblasr/MappingParameters.h with its defaults (:207-381) after
    MakeSane (:501-503 turns bandSize 0 into 16 when guided alignment is on)."""
    insertion: int = 4
    deletion: int = 5
    indel: int = 5
    match: int = -5          # SMRTDistanceMatrix diagonal after Blasr.cpp:910-920
    mismatch: int = 6
    affineAlign: bool = False
    affineOpen: int = 10
    affineExtend: int = 0
    guidedAlignBandSize: int = 10
    bandSize: int = 16
    useGuidedAlign: bool = True
    doGlobalAlignment: bool = False
    refineAlignments: bool = True
    sortRefinedAlignments: bool = True
    maxScore: int = -200
    nCandidates: int = 10    # :243
    nBest: int = 10          # :242
    minPctSimilarity: float = 0.0   # :299
    minFractionToBeConsideredOverlapping: float = 0.75   # :212
    maxBandSize: int = 250   # libcpp's MAX_BAND_SIZE (ctest/concordant.t:23 mentions it)
    tieBreak: int = _lib.TIE_DIAG_LEFT_UP
    sdpTupleSize: int = 11   # :222
    sdpIns: int = 5          # :220
    sdpDel: int = 10         # :221
    sdpMaxFragments: int = 1024   # not a BLASR option: the device's budget of maximal matches per candidate
    sdpBypassThreshold: float = 1000000.0   # :340 (0.75 in the -sdpbypass preset, :721)
    scoreMatrix: Optional[list] = None     # 5x5, -scoreMatrix (Blasr.cpp:922-937)
    ignoreQualities: bool = True           # :285; -useQuality turns it off
    refineBetweenAnchorsOnly: bool = False  # -rbao (iblasr/RegisterBlasrOptions.h; MappingParameters.h)


@dataclass
class AlignmentCandidate:
    """The T_AlignmentCandidate fields RefineAlignment reads and writes (:678-772)."""
    tAlignedSeq: bytes                      # target window (already reverse-complemented for rev-strand hits)
    tAlignedSeqPos: int = 0
    qAlignedSeqPos: int = 0
    qPos: int = 0
    tPos: int = 0
    blocks: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=BLOCK_DT))
    tStrand: int = 0
    tIndex: int = 0                         # index of the contig the window lies in (seqdb), compared at BlasrUtilsImpl.hpp:552
    score: int = 0
    nCells: int = 0
    nMatch: int = 0
    nMismatch: int = 0
    nIns: int = 0
    nDel: int = 0
    pctSimilarity: float = 0.0
    qAlignedSeq: bytes = b""
    tAlignedSeqLength: int = 0
    qAlignedSeqLength: int = 0
    status: int = 0
    probScore: float = 0.0
    mapQV: int = 0
    # filled by copy_back from the pair result: the aligned stretch of the read / of the target window
    # (GetQInterval / the -m 4 columns) and the number of blocks, which are all an interval printer and
    # the mapping qualities need when the blocks themselves were not asked for
    qInterval: Optional[tuple] = None
    tInterval: Optional[tuple] = None
    nBlocks: int = 0

    def n_blocks(self) -> int:
        return max(len(self.blocks), self.nBlocks)


# RefineAlignment never assigns affineOpen / affineExtend of its two DistanceMatrixScoreFunction
# objects (only idsScoreFn receives params.affineOpen / affineExtend, :643-644; StoreMapQVs does
# assign them, :120-122, under the comment "bug 24363"), so both scorers run with whatever
# libcpp's constructor leaves there.  libcpp is not in the reference tree: the value is UNPINNED,
# believed to be 0 (a base-class constructor that zeroes every cost).  One place to change it:
DIST_SCORER_AFFINE_OPEN = 0        # distScoreFn.affineOpen  in RefineAlignment (DP of -affineAlign without QVs)
DIST_SCORER_AFFINE_EXTEND = 0      # distScoreFn.affineExtend
STATS_SCORER_AFFINE_OPEN = 0       # distScoreFn2.affineOpen  (ComputeAlignmentStats(..., params.affineAlign), :733-736)
STATS_SCORER_AFFINE_EXTEND = 0     # distScoreFn2.affineExtend


def scoring_from_params(p: MappingParameters, ids: bool = False, align_type: int = _lib.ALIGN_GLOBAL):
    """distScoreFn(SMRTDistanceMatrix, params.deletion, params.insertion) read as ctor(ins, del)
    (:630-631) and distScoreFn2(indel, indel) (:633-634); band per :706-731.  The affine costs of
    both scorers are the constructor defaults above, NOT params.affineOpen / affineExtend.
    ids: the DP runs with idsScoreFn, the one scorer that does receive params.affineOpen / affineExtend
    (:643-644); its per-base costs travel beside this as a profile."""
    if p.doGlobalAlignment:
        raise ValueError("the -global branch has its own scorer and band: use scoring_global() / refine_global()")
    mode = _lib.MODE_AFFINE if p.affineAlign else _lib.MODE_LINEAR
    band = p.bandSize if p.affineAlign else p.guidedAlignBandSize
    return make_scoring(match=p.match, mismatch=p.mismatch, ins=p.deletion, dele=p.insertion,
                        indel=p.indel, band=band, max_band=p.maxBandSize, mode=mode,
                        tie_break=p.tieBreak, affine_open=p.affineOpen if ids else DIST_SCORER_AFFINE_OPEN,
                        affine_extend=p.affineExtend if ids else DIST_SCORER_AFFINE_EXTEND,
                        stats_affine=1 if p.affineAlign else 0, stats_affine_open=STATS_SCORER_AFFINE_OPEN,
                        stats_affine_extend=STATS_SCORER_AFFINE_EXTEND, sub=p.scoreMatrix, align_type=align_type)


def compute_drift(blocks):
    """ComputeDrift(alignmentCandidate) (:654, :793; libcpp AlignmentUtils, not in the tree): the
    largest distance of a block from the diagonal of the first block.  UNPINNED; restated from the
    way both call sites use it (as the half-width of a diagonal band that must hold the guide)."""
    if len(blocks) == 0:
        return 0
    d0 = int(blocks[0]["t_pos"]) - int(blocks[0]["q_pos"])
    return max(abs(int(b["t_pos"]) - int(b["q_pos"]) - d0) for b in blocks)


def scoring_global(p: MappingParameters, k: int):
    """KBandAlign(subread, tAlignedSeq, SMRTDistanceMatrix, params.insertion, params.deletion, drift,
    ..., idsScoreFn, Global) of the -global branch (:657-661), rescored by distScoreFn2 WITHOUT the
    affine flag (:663-666).  The IDS scorer adds per-base quality terms to insertion / deletion /
    substitution; for a read without quality values what is left of it is the matrix and the two
    explicit gap costs, which is what runs here (the QV terms belong to the QV-aware scorers)."""
    return make_scoring(match=p.match, mismatch=p.mismatch, ins=p.insertion, dele=p.deletion, indel=p.indel,
                        mode=_lib.MODE_KBAND, kband_k=int(k), tie_break=p.tieBreak, sub=p.scoreMatrix)


def build_batch(read: bytes, cands: List[AlignmentCandidate]):
    """Slicing rules of :693-703.  Every candidate views the same forward read; the target is
    copied from its window.  Candidates with no blocks stay in the batch with n_blk = 0 and come
    back untouched (status EMPTY_GUIDE), like :681."""
    n = len(cands)
    q_arena = np.frombuffer(read, dtype=np.uint8)
    q_off = np.zeros(n, np.uint64); q_len = np.zeros(n, np.uint32)
    t_off = np.zeros(n, np.uint64); t_len = np.zeros(n, np.uint32)
    b_off = np.zeros(n, np.uint64); n_blk = np.zeros(n, np.uint32)
    t_parts, b_parts, tpos, bpos = [], [], 0, 0
    for i, c in enumerate(cands):
        nb = len(c.blocks)
        if nb:
            last = c.blocks[nb - 1]
            tl = int(last["t_pos"]) + int(last["length"]) - int(c.blocks[0]["t_pos"])
            ql = int(last["q_pos"]) + int(last["length"])
            q_off[i] = c.qAlignedSeqPos + c.qPos; q_len[i] = ql
            t = np.frombuffer(c.tAlignedSeq, dtype=np.uint8)[c.tPos:c.tPos + tl]
            t_off[i] = tpos; t_len[i] = len(t); t_parts.append(t); tpos += len(t)
            b_off[i] = bpos; n_blk[i] = nb; b_parts.append(np.ascontiguousarray(c.blocks, dtype=BLOCK_DT)); bpos += nb
    return {"n_pairs": n, "q_arena": q_arena, "q_off": q_off, "q_len": q_len,
            "t_arena": np.concatenate(t_parts) if t_parts else np.zeros(1, np.uint8),
            "t_off": t_off, "t_len": t_len,
            "blocks": np.concatenate(b_parts) if b_parts else np.zeros(1, BLOCK_DT),
            "blk_off": b_off, "n_blk": n_blk}


def copy_back(read: bytes, cands: List[AlignmentCandidate], batch, pairs, blocks):
    """:733-772 for every candidate the kernels aligned.  blocks None: the batch ran without a block
    arena (interval output); the candidate then has no blocks, only their number and the intervals."""
    for i, (c, r) in enumerate(zip(cands, pairs)):
        c.status = int(r["status"])
        if c.status != _lib.PAIR_OK:
            continue
        if blocks is None:
            c.blocks = np.zeros(0, dtype=BLOCK_DT)
        else:
            c.blocks = blocks[int(r["blk_off"]):int(r["blk_off"]) + int(r["n_blk"])].copy()
        c.nBlocks = int(r["n_blk"])
        c.nMatch, c.nMismatch = int(r["n_match"]), int(r["n_mismatch"])
        c.nIns, c.nDel = int(r["n_ins"]), int(r["n_del"])
        cols = c.nMatch + c.nMismatch + c.nIns + c.nDel
        c.pctSimilarity = (c.nMatch * 2.0) / (2 * cols) * 100.0 if cols else 0.0
        c.score, c.nCells = int(r["score"]), int(r["ncells"])
        to, tl = int(batch["t_off"][i]), int(batch["t_len"][i])
        qo, ql = int(batch["q_off"][i]), int(batch["q_len"][i])
        c.tAlignedSeq = batch["t_arena"][to:to + tl].tobytes()
        c.qAlignedSeq = read[qo:qo + ql]
        c.tAlignedSeqPos += c.tPos
        c.qAlignedSeqPos += c.qPos
        c.tPos, c.qPos = int(r["t_pos"]), int(r["q_pos"])
        c.tAlignedSeqLength, c.qAlignedSeqLength = tl, ql
        c.qInterval = (c.qAlignedSeqPos + int(r["q_start"]), c.qAlignedSeqPos + int(r["q_end"]))
        c.tInterval = (c.tAlignedSeqPos + int(r["t_start"]), c.tAlignedSeqPos + int(r["t_end"]))


def SDPAlignCandidates(read: bytes, cands: List[AlignmentCandidate], params: MappingParameters, ctx):
    """The guide of every candidate of a read in one call: what AlignIntervals does per candidate with
    SDPAlign(alignment->qAlignedSeq, alignment->tAlignedSeq, distScoreFn, sdpTupleSize, params.sdpIns,
    params.sdpDel, params.indelRate*3, *alignment, mappingBuffers, Local, ...)
    (/root/reference/iblasr/BlasrAlignImpl.hpp:980-987).  The query is the read from qAlignedSeqPos
    on (qAlignedSeqLength bases, the rest of the read when that is 0), the target the candidate's
    window.  On return blocks are relative to the first block and qPos / tPos give its start, the
    convention of :951-964; a candidate without a single match of sdpTupleSize bases keeps no blocks."""
    n = len(cands)
    if n == 0:
        return cands
    q_arena = np.frombuffer(read, dtype=np.uint8)
    q_off = np.array([c.qAlignedSeqPos for c in cands], np.uint64)
    q_len = np.array([c.qAlignedSeqLength or len(read) - c.qAlignedSeqPos for c in cands], np.uint32)
    t_len = np.array([len(c.tAlignedSeq) for c in cands], np.uint32)
    t_off = np.zeros(n, np.uint64); t_off[1:] = np.cumsum(t_len[:-1].astype(np.uint64))
    t_arena = np.frombuffer(b"".join(c.tAlignedSeq for c in cands) or b"N", dtype=np.uint8)
    batch = {"n_pairs": n, "q_arena": q_arena, "q_off": q_off, "q_len": q_len, "t_arena": t_arena, "t_off": t_off, "t_len": t_len}
    sp = make_sdp_params(tuple_size=params.sdpTupleSize, ins=params.sdpIns, dele=params.sdpDel, match=-params.match,
                         max_fragments=params.sdpMaxFragments)
    status, score, n_blk, off, blocks, rc = ctx.sdp_host(batch, sp, blocks_capacity=n * params.sdpMaxFragments)
    for i, c in enumerate(cands):
        c.status = int(status[i])
        b = blocks[int(off[i]):int(off[i]) + int(n_blk[i])].copy() if c.status == _lib.PAIR_OK else np.zeros(0, dtype=BLOCK_DT)
        c.qPos = c.tPos = 0
        if len(b):
            c.qPos, c.tPos = int(b[0]["q_pos"]), int(b[0]["t_pos"])
            b["q_pos"] -= np.uint32(c.qPos); b["t_pos"] -= np.uint32(c.tPos)
        c.blocks = b
        c.score = -int(score[i])              # lower is better, like every score of the reference
    return cands


def AlignAndRefineCandidates(read: bytes, cands: List[AlignmentCandidate], params: MappingParameters, ctx, want_blocks: bool = True):
    """SDPAlign (AlignIntervals, /root/reference/iblasr/BlasrAlignImpl.hpp:980-987) and RefineAlignment
    (/root/reference/iblasr/BlasrUtilsImpl.hpp:905-923) for every candidate of a read in ONE submit
    (Context.set_guide_producer): the batch carries the read and the candidates' windows, the guides are made
    on the device and refined where they lie; no guide travels.  The candidates end up as
    SDPAlignCandidates followed by RefineAlignments leaves them; a candidate without a guide keeps no
    blocks and gets the producer's reason as its status.  want_blocks False: interval output."""
    n = len(cands)
    if n == 0:
        return cands
    q_arena = np.frombuffer(read, dtype=np.uint8)
    q_off = np.array([c.qAlignedSeqPos for c in cands], np.uint64)
    q_len = np.array([c.qAlignedSeqLength or len(read) - c.qAlignedSeqPos for c in cands], np.uint32)
    t_len = np.array([len(c.tAlignedSeq) for c in cands], np.uint32)
    t_off = np.zeros(n, np.uint64); t_off[1:] = np.cumsum(t_len[:-1].astype(np.uint64))
    t_arena = np.frombuffer(b"".join(c.tAlignedSeq for c in cands) or b"N", dtype=np.uint8)
    batch = {"n_pairs": n, "q_arena": q_arena, "q_off": q_off, "q_len": q_len, "t_arena": t_arena, "t_off": t_off, "t_len": t_len}
    ctx.set_scoring(scoring_from_params(params))
    ctx.set_guide_producer(make_sdp_params(tuple_size=params.sdpTupleSize, ins=params.sdpIns, dele=params.sdpDel, match=-params.match,
                                           max_fragments=params.sdpMaxFragments))
    try:
        if want_blocks:
            pairs, blocks, rc = ctx.refine_host(batch)
        else:
            (pairs, rc), blocks = ctx.refine_host_intervals(batch), None
    finally:
        ctx.set_guide_producer(None)
    for i, (c, r) in enumerate(zip(cands, pairs)):
        c.status = int(r["status"])
        c.blocks = np.zeros(0, dtype=BLOCK_DT)
        c.nBlocks = 0
        c.qPos = c.tPos = 0
        if c.status != _lib.PAIR_OK:
            continue
        if blocks is not None:
            c.blocks = blocks[int(r["blk_off"]):int(r["blk_off"]) + int(r["n_blk"])].copy()
        c.nBlocks = int(r["n_blk"])
        c.nMatch, c.nMismatch = int(r["n_match"]), int(r["n_mismatch"])
        c.nIns, c.nDel = int(r["n_ins"]), int(r["n_del"])
        cols = c.nMatch + c.nMismatch + c.nIns + c.nDel
        c.pctSimilarity = (c.nMatch * 2.0) / (2 * cols) * 100.0 if cols else 0.0
        c.score, c.nCells = int(r["score"]), int(r["ncells"])
        # the refined stretch starts at the chain's first anchor: that becomes the start of the aligned sequences
        qp, tp, rows, ncol = int(r["q_pos"]), int(r["t_pos"]), int(r["rows"]), int(r["cols"])
        c.qInterval = (c.qAlignedSeqPos + int(r["q_start"]), c.qAlignedSeqPos + int(r["q_end"]))
        c.tInterval = (c.tAlignedSeqPos + int(r["t_start"]), c.tAlignedSeqPos + int(r["t_end"]))
        qo = int(q_off[i]) + qp
        c.qAlignedSeq = read[qo:qo + rows]
        c.tAlignedSeq = c.tAlignedSeq[tp:tp + ncol]
        c.qAlignedSeqPos += qp
        c.tAlignedSeqPos += tp
        c.qAlignedSeqLength, c.tAlignedSeqLength = rows, ncol
    if params.sortRefinedAlignments:
        cands.sort(key=lambda c: c.score)
    return cands


def GuidesFromAnchorsOrSDP(read: bytes, cands: List[AlignmentCandidate], anchors: List[list], params: MappingParameters, ctx,
                           subreadLength: Optional[int] = None):
    """AlignIntervals' choice of the guide (/root/reference/iblasr/BlasrAlignImpl.hpp:768-1004): when
    the anchors of a candidate (its interval's exact matches, (q, t, l) in read / genome-window
    coordinates) cover at least params.sdpBypassThreshold of the subread they are the guide
    themselves (:992-1004: block = anchor - aligned-sequence position), otherwise the guide comes
    from SDPAlign over the candidate's two sequences (:980-987).  The default threshold (1e6)
    sends every candidate to SDPAlign.  With params.refineBetweenAnchorsOnly (-rbao) the candidates that
    need SDP get it between their anchors only (GuidesBetweenAnchors, :785-978)."""
    n_sub = subreadLength if subreadLength is not None else len(read)
    if params.refineBetweenAnchorsOnly:
        sparse = [(c, a) for c, a in zip(cands, anchors) if not (n_sub and sum(int(m[2]) for m in a) / n_sub >= params.sdpBypassThreshold)]
        dense = [(c, a) for c, a in zip(cands, anchors) if n_sub and sum(int(m[2]) for m in a) / n_sub >= params.sdpBypassThreshold]
        if sparse:
            GuidesBetweenAnchors(read, [c for c, _ in sparse], [a for _, a in sparse], params, ctx)
        cands_, anchors_ = [c for c, _ in dense], [a for _, a in dense]
    else:
        cands_, anchors_ = cands, anchors
    todo = []
    for c, a in zip(cands_, anchors_):
        covered = sum(int(m[2]) for m in a)
        if n_sub and covered / n_sub >= params.sdpBypassThreshold:
            b = np.zeros(len(a), dtype=BLOCK_DT)
            for i, (q, t, l) in enumerate(a):
                b[i] = (int(q) - c.qAlignedSeqPos, int(t) - c.tAlignedSeqPos, int(l))
            c.blocks = b
            c.status = _lib.PAIR_OK if len(b) else _lib.PAIR_EMPTY_GUIDE
        else:
            todo.append(c)
    if todo:
        SDPAlignCandidates(read, todo, params, ctx)
    return cands


def GuidesBetweenAnchors(read: bytes, cands: List[AlignmentCandidate], anchors: List[list], params: MappingParameters, ctx):
    """The -rbao branch of AlignIntervals (/root/reference/iblasr/BlasrAlignImpl.hpp:785-978): the guide of a candidate is its
    anchors (exact matches (q, t, l) in read / genome coordinates, in order, forward strand; the strand flip of :828-851 is the
    caller's) with, between two consecutive anchors that leave a gap in BOTH sequences, the chain SDPAlign finds inside that
    gap (:902-908: the sub-sequences between the anchors, params.sdpTupleSize, sdpIns, sdpDel).  All gaps of all candidates
    go through one gba_sdp call.  As in the reference an anchor is taken into the guide together with the gap behind it:
    an anchor that is followed by no gap in the query or none in the target (tGap <= 0 or qGap <= 0, :877) is left out;
    the last anchor always goes in (:938-949).  Blocks end up relative to the first one, qPos / tPos at its start (:951-964).
    SDPAlign's Global flag for these calls is libcpp's and unpinned: the chain is the same best chain gba_sdp returns."""
    jobs = []          # (candidate index, anchor index, q start, t start, q gap, t gap) in read / window coordinates
    for ci, (c, a) in enumerate(zip(cands, anchors)):
        m = [(int(q) - c.qAlignedSeqPos, int(t) - c.tAlignedSeqPos, int(l)) for q, t, l in a]      # relative to the aligned sequences (:822-827)
        for k in range(len(m) - 1):
            t_gap = m[k + 1][1] - (m[k][1] + m[k][2])
            q_gap = m[k + 1][0] - (m[k][0] + m[k][2])
            if t_gap > 0 and q_gap > 0:
                jobs.append((ci, k, m[k][0] + m[k][2], m[k][1] + m[k][2], q_gap, t_gap))
    found = {}
    if jobs:
        # one batch: the query arena is the read, the target arena the candidates' windows one after the other
        t_base, acc = [], 0
        for c in cands:
            t_base.append(acc); acc += len(c.tAlignedSeq)
        batch = {"n_pairs": len(jobs), "q_arena": np.frombuffer(read, dtype=np.uint8),
                 "t_arena": np.frombuffer(b"".join(c.tAlignedSeq for c in cands) or b"N", dtype=np.uint8),
                 "q_off": np.array([cands[j[0]].qAlignedSeqPos + j[2] for j in jobs], np.uint64), "q_len": np.array([j[4] for j in jobs], np.uint32),
                 "t_off": np.array([t_base[j[0]] + j[3] for j in jobs], np.uint64), "t_len": np.array([j[5] for j in jobs], np.uint32)}
        sp = make_sdp_params(tuple_size=params.sdpTupleSize, ins=params.sdpIns, dele=params.sdpDel, match=-params.match,
                             max_fragments=params.sdpMaxFragments)
        status, score, n_blk, off, blocks, rc = ctx.sdp_host(batch, sp, blocks_capacity=int(sum(min(j[4], j[5]) for j in jobs)) + len(jobs))
        for x, j in enumerate(jobs):
            if int(status[x]) == _lib.PAIR_OK and int(n_blk[x]):
                b = blocks[int(off[x]):int(off[x]) + int(n_blk[x])].copy()
                b["q_pos"] += np.uint32(j[2]); b["t_pos"] += np.uint32(j[3])          # :921-923
                found[(j[0], j[1])] = b
    for ci, (c, a) in enumerate(zip(cands, anchors)):
        m = [(int(q) - c.qAlignedSeqPos, int(t) - c.tAlignedSeqPos, int(l)) for q, t, l in a]
        parts = []
        for k in range(len(m) - 1):
            t_gap = m[k + 1][1] - (m[k][1] + m[k][2])
            q_gap = m[k + 1][0] - (m[k][0] + m[k][2])
            if t_gap > 0 and q_gap > 0:
                parts.append(np.array([m[k]], dtype=BLOCK_DT))
                if (ci, k) in found:
                    parts.append(found[(ci, k)])
        if m:
            parts.append(np.array([m[-1]], dtype=BLOCK_DT))
        b = np.concatenate(parts) if parts else np.zeros(0, dtype=BLOCK_DT)
        c.qPos = c.tPos = 0
        if len(b):
            c.qPos, c.tPos = int(b[0]["q_pos"]), int(b[0]["t_pos"])
            b["q_pos"] -= np.uint32(c.qPos); b["t_pos"] -= np.uint32(c.tPos)
        c.blocks = b
        c.status = _lib.PAIR_OK if len(b) else _lib.PAIR_EMPTY_GUIDE
    return cands


def GuidedAlignFromSequences(read: bytes, cands: List[AlignmentCandidate], params: MappingParameters, ctx,
                             bandSize: int = 12, sdpTupleSize: int = 6, qv=None):
    """The GuidedAlign overload that takes only the two sequences: it chains matches of sdpTupleSize
    bases itself and runs the banded DP around that chain.  AlignSubreadToAlignmentTarget uses it to
    align every subread of a ZMW to the window its CCS (or longest subread) mapped to:
    GuidedAlign(subread, alignedRefSequence, idsScoreFn, 12, params.sdpIns, params.sdpDel,
    params.indelRate, mappingBuffers, exploded, Local, computeProbIsFalse, 6)
    (/root/reference/iblasr/BlasrAlignImpl.hpp:1418-1421), then ComputeAlignmentStats with
    distScoreFn2 (:1448-1450).  Here: SDPAlignCandidates with the short tuple, then the refine stage
    with band `bandSize`, all candidates of the read in one batch each:
    the DP region runs from the chain's first block to its last and the Local end rule (gba.h, GBA_ALIGN_LOCAL) lets the
    alignment begin and end anywhere inside it; libcpp's own Local handling is not in the reference tree (unpinned).  qv:
    the read's quality values; with them (and params.ignoreQualities off) the DP runs on the IDS scorer's costs."""
    import dataclasses
    p = dataclasses.replace(params, sdpTupleSize=sdpTupleSize, guidedAlignBandSize=bandSize, useGuidedAlign=True,
                            doGlobalAlignment=False, refineAlignments=True, affineAlign=False)
    SDPAlignCandidates(read, cands, p, ctx)
    return RefineAlignments(read, cands, p, ctx, qv=qv, align_type=_lib.ALIGN_LOCAL)


def RefineAlignments(read: bytes, cands: List[AlignmentCandidate], params: MappingParameters, ctx, want_blocks: bool = True, qv=None,
                     align_type: int = _lib.ALIGN_GLOBAL):
    """Batched RefineAlignments (:905-923): all candidates of the read in one submit.
    want_blocks False: for interval / summary output (printFormat Interval, SummaryPrint, :992-1000), which
    prints only what the pair results hold: no block leaves the device.
    qv: the read's QualityValues.  With params.ignoreQualities off and meaningful quality values the guided DP runs with
    the IDS scorer (GuidedAlign(qSeq, tSeq, alignmentCandidate, idsScoreFn, params.guidedAlignBandSize, ...), :705-717):
    its costs travel as gba_batch.q_profile (ids_profile).  With params.affineAlign that is AffineGuidedAlign with the IDS
    scorer (:707-710): per-base substitution / insertion / deletion costs (within +-63) and params.affineOpen / affineExtend."""
    if not params.refineAlignments or not cands:
        return cands
    use_qv = qv is not None and not params.ignoreQualities and ReadHasMeaningfulQualityValues(qv)
    if params.doGlobalAlignment:
        if use_qv:
            refine_global_qv(read, cands, params, ctx, qv)
        else:
            refine_global(read, cands, params, ctx)
        if params.sortRefinedAlignments:
            cands.sort(key=lambda c: c.score)
        return cands
    if not params.useGuidedAlign:
        refine_legacy(read, cands, params, ctx, qv if use_qv else None)
        if params.sortRefinedAlignments:
            cands.sort(key=lambda c: c.score)
        return cands
    ctx.set_scoring(scoring_from_params(params, ids=use_qv, align_type=align_type))
    batch = build_batch(read, cands)
    if use_qv:
        batch["q_profile"] = ids_profile(read, qv, params)
    if want_blocks:
        pairs, blocks, rc = ctx.refine_host(batch)
    else:
        (pairs, rc), blocks = ctx.refine_host_intervals(batch), None
    copy_back(read, cands, batch, pairs, blocks)
    if params.sortRefinedAlignments:
        cands.sort(key=lambda c: c.score)        # stable, ascending: lower is better (:920-922)
    return cands


def remove_end_gaps(c: AlignmentCandidate):
    """Alignment::RemoveEndGaps() as :662 uses it: the alignment starts at its first block and ends
    at its last one; a gap in front of the first block moves into tPos / qPos."""
    if len(c.blocks) == 0:
        return
    dq, dt = int(c.blocks[0]["q_pos"]), int(c.blocks[0]["t_pos"])
    if dq or dt:
        c.blocks = c.blocks.copy()
        c.blocks["q_pos"] -= dq
        c.blocks["t_pos"] -= dt
        c.qPos += dq
        c.tPos += dt


def refine_global(read: bytes, cands: List[AlignmentCandidate], params: MappingParameters, ctx,
                  subread_start: int = 0, subread_length: Optional[int] = None):
    """The -global branch of RefineAlignment (:648-675) for all candidates of a read: the WHOLE
    subread against the WHOLE tAlignedSeq in a diagonal band of half-width k = ComputeDrift(candidate),
    RemoveEndGaps, rescoring with distScoreFn2, then blocks / tPos / qPos (+ SubreadStart) / score are
    copied back; nothing else of the candidate changes.  Candidates whose end cell lies outside their
    band keep their status (END_OUT_OF_BAND) and are not touched.  One submit per distinct k."""
    if subread_length is None:
        subread_length = len(read) - subread_start
    sub = read[subread_start:subread_start + subread_length]
    by_k = {}
    for i, c in enumerate(cands):
        by_k.setdefault(compute_drift(c.blocks), []).append(i)
    for k, idx in sorted(by_k.items()):
        ctx.set_scoring(scoring_global(params, k))
        n = len(idx)
        t_parts = [np.frombuffer(cands[i].tAlignedSeq, dtype=np.uint8) for i in idx]
        t_len = np.array([len(t) for t in t_parts], np.uint32)
        t_off = np.zeros(n, np.uint64)
        t_off[1:] = np.cumsum(t_len[:-1], dtype=np.uint64)
        batch = {"n_pairs": n, "q_arena": np.frombuffer(sub, dtype=np.uint8) if len(sub) else np.zeros(1, np.uint8),
                 "q_off": np.zeros(n, np.uint64), "q_len": np.full(n, len(sub), np.uint32),
                 "t_arena": np.concatenate(t_parts) if t_len.sum() else np.zeros(1, np.uint8), "t_off": t_off, "t_len": t_len,
                 "blocks": np.zeros(1, BLOCK_DT), "blk_off": np.zeros(n, np.uint64), "n_blk": np.zeros(n, np.uint32)}
        pairs, blocks, rc = ctx.refine_host(batch)
        for i, r in zip(idx, pairs):
            c = cands[i]
            c.status = int(r["status"])
            if c.status != _lib.PAIR_OK:
                continue
            c.blocks = blocks[int(r["blk_off"]):int(r["blk_off"]) + int(r["n_blk"])].copy()
            c.tPos, c.qPos = int(r["t_pos"]), int(r["q_pos"])
            remove_end_gaps(c)
            c.qPos += subread_start                      # :672
            c.score = int(r["score"])                    # :673
    return cands


def RemoveLowQualityAlignments(cands: List[AlignmentCandidate], params: MappingParameters):
    """Among the first nCandidates candidates, drop from the first one that has no blocks or scores above
    maxScore to the end (iblasr/BlasrUtilsImpl.hpp:485-509), as MapRead does right after refining."""
    for i in range(min(params.nCandidates, len(cands))):
        c = cands[i]
        if c.n_blocks() == 0 or c.score > params.maxScore:
            del cands[i:]
            break
    return cands


def genomic_t_interval(c: AlignmentCandidate):
    """GenomicTBegin() / GenomicTEnd() of a candidate (libcpp's Alignment, not in the reference tree: UNPINNED): the
    stretch of its target strand between the first aligned base and the end of the last block,
    tAlignedSeqPos + tPos (+ TEnd())."""
    if c.n_blocks() == 0:
        b = c.tAlignedSeqPos + c.tPos
        return b, b
    if len(c.blocks) == 0:                      # refined without a block arena: the interval came with the pair result
        return c.tInterval
    last = c.blocks[len(c.blocks) - 1]
    b = c.tAlignedSeqPos + c.tPos
    return b, b + int(last["t_pos"]) + int(last["length"])


def RemoveOverlappingAlignments(cands: List[AlignmentCandidate], params: MappingParameters):
    """RemoveOverlappingAlignments (iblasr/BlasrUtilsImpl.hpp:523-605), what MapRead runs after the refine and the
    quality filter (iblasr/BlasrAlignImpl.hpp:382): of two candidates on the same contig whose target stretches
    contain one another, the contained one goes when its score is no better (j inside i and score_i <= score_j drops
    j; i inside j and score_j <= score_i drops i); a candidate below minPctSimilarity is not used as the outer one;
    the survivors keep their order.  Returns the number left."""
    n = len(cands)
    contained = [False] * n
    for i in range(max(n - 1, 0)):
        a = cands[i]
        if a.pctSimilarity < params.minPctSimilarity:
            continue
        ab, ae = genomic_t_interval(a)
        for j in range(i + 1, n):
            if contained[j] or a.tIndex != cands[j].tIndex:
                continue
            bb, be = genomic_t_interval(cands[j])
            if ab <= bb and ae >= be:
                if a.score <= cands[j].score:
                    contained[j] = True
            elif bb <= ab and be >= ae:
                if cands[j].score <= a.score:
                    contained[i] = True
    cands[:] = [c for c, gone in zip(cands, contained) if not gone]
    return len(cands)


# ------------------------------------------------------------------------------------------------
# The fixed-band aligners around the guided path (gba_kband): quality-value scorers, the -global
# branch with them, the legacy k-band refine, and the end extension of -extend.

@dataclass
class QualityValues:
    """The per-base fields of a FASTQSequence / SMRTSequence that the QV-aware scorers read
    (bytes or uint8 arrays as long as the read; None = absent, like an Empty() QV vector)."""
    qual: Optional[bytes] = None
    insertionQV: Optional[bytes] = None
    deletionQV: Optional[bytes] = None
    deletionTag: Optional[bytes] = None
    substitutionQV: Optional[bytes] = None
    substitutionTag: Optional[bytes] = None


_ACGTN = b"ACGTN"
_CODE = np.full(256, 4, np.int64)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
    _CODE[_c + 32] = _i


def _matrix(p: MappingParameters):
    if p.scoreMatrix is not None:
        return np.asarray(p.scoreMatrix, np.int64)
    m = np.full((5, 5), p.mismatch, np.int64)
    m[np.arange(4), np.arange(4)] = p.match
    m[4, :] = 0
    m[:, 4] = 0
    return m


def ids_profile(read: bytes, qv: QualityValues, p: MappingParameters, substitutionPrior: int = 20, globalDeletionPrior: int = 13):
    """IDSScoreFunction<DNASequence, FASTQSequence> as RefineAlignment sets it up
    (/root/reference/iblasr/BlasrUtilsImpl.hpp:636-646: ins = params.insertion, del = params.deletion,
    substitutionPrior, globalDeletionPrior, InitializeScoreMatrix(SMRTDistanceMatrix); defaults
    iblasr/MappingParameters.h:350-351), evaluated for every query base into the cost profile gba_kband
    takes.  The class is libcpp code and not in the reference tree; UNPINNED, restated as
      Match(t, q_i)     = matrix[q_i][q_i] when the bases agree; substitutionQV[i] when the target base is
                          the read's substitutionTag[i]; substitutionPrior otherwise;
      Insertion(q_i)    = insertionQV[i] (ins without insertion QVs);
      Deletion(t, q_i)  = deletionQV[i] when the target base is the read's deletionTag[i];
                          globalDeletionPrior otherwise (del without deletion QVs / tags).
    Every value is clamped to the +-1023 the kernels take."""
    n = len(read)
    q = np.frombuffer(read, np.uint8)
    qc = _CODE[q]
    m = _matrix(p)
    prof = np.zeros((n, 11), np.int64)
    arr = lambda b: np.frombuffer(b, np.uint8).astype(np.int64) if b is not None else None
    sq, st = arr(qv.substitutionQV), arr(qv.substitutionTag)
    iq, dq, dt = arr(qv.insertionQV), arr(qv.deletionQV), arr(qv.deletionTag)
    for c in range(5):
        col = np.full(n, substitutionPrior, np.int64)
        if sq is not None and st is not None:
            col = np.where(_CODE[st] == c, sq, col) if c < 4 else col
        col = np.where(qc == c, m[qc, qc], col)
        prof[:, c] = col
        if dq is not None and dt is not None:
            prof[:, 6 + c] = np.where((_CODE[dt] == c) & (dt != ord("N")), dq, globalDeletionPrior) if c < 4 else globalDeletionPrior
        else:
            prof[:, 6 + c] = p.deletion
    prof[:, 5] = iq if iq is not None else p.insertion
    return np.clip(prof, -1023, 1023).astype(np.int16).reshape(-1)


def qv_profile(read: bytes, qv: QualityValues, p: MappingParameters):
    """QualityValueScoreFunction with ins = del = params.indel (:639-640; BlasrAlignImpl.hpp:1241-1243):
    UNPINNED, restated as Match = 0 for equal bases and the base's quality value otherwise, constant gaps."""
    n = len(read)
    qc = _CODE[np.frombuffer(read, np.uint8)]
    qual = np.frombuffer(qv.qual, np.uint8).astype(np.int64) if qv.qual is not None else np.full(n, p.mismatch, np.int64)
    prof = np.zeros((n, 11), np.int64)
    for c in range(5):
        prof[:, c] = np.where(qc == c, 0, qual)
    prof[:, 5] = p.indel
    prof[:, 6:] = p.indel
    return np.clip(prof, -1023, 1023).astype(np.int16).reshape(-1)


def ReadHasMeaningfulQualityValues(qv: Optional[QualityValues]):
    """The test PairwiseLocalAlign makes before it trusts quality values (BlasrAlignImpl.hpp:1257): some
    quality value of the read differs from 0."""
    return qv is not None and qv.qual is not None and bool(np.frombuffer(qv.qual, np.uint8).any())


def _kband_params(p: MappingParameters, align_type, gap_model=_lib.KB_LINEAR, ins=None, dele=None, max_drops=10):
    from .api import make_kband_params
    return make_kband_params(align_type=align_type, gap_model=gap_model, ins=p.insertion if ins is None else ins,
                             dele=p.deletion if dele is None else dele, indel=p.indel, sub=_matrix(p).tolist(),
                             tie_break=p.tieBreak, max_drops=max_drops)


def _kband_batch(seq_pairs, ks, profile_parts=None, reverse=None):
    n = len(seq_pairs)
    q_len = np.array([len(q) for q, _ in seq_pairs], np.uint32)
    t_len = np.array([len(t) for _, t in seq_pairs], np.uint32)
    q_off = np.zeros(n, np.uint64); t_off = np.zeros(n, np.uint64)
    q_off[1:] = np.cumsum(q_len[:-1].astype(np.uint64)); t_off[1:] = np.cumsum(t_len[:-1].astype(np.uint64))
    b = {"n_pairs": n, "q_arena": np.frombuffer(b"".join(bytes(q) for q, _ in seq_pairs) or b"N", np.uint8),
         "t_arena": np.frombuffer(b"".join(bytes(t) for _, t in seq_pairs) or b"N", np.uint8),
         "q_off": q_off, "t_off": t_off, "q_len": q_len, "t_len": t_len, "k": np.asarray(ks, np.int32)}
    if profile_parts is not None:
        prof = np.concatenate(profile_parts) if sum(len(x) for x in profile_parts) else np.zeros(11, np.int16)
        b["profile"] = np.ascontiguousarray(prof, np.int16) if len(prof) == len(b["q_arena"]) * 11 else np.zeros(len(b["q_arena"]) * 11, np.int16)
    if reverse is not None:
        b["reverse"] = np.asarray(reverse, np.uint8)
    return b


def _stats_into(c: AlignmentCandidate, r):
    c.nMatch, c.nMismatch, c.nIns, c.nDel = int(r["n_match"]), int(r["n_mismatch"]), int(r["n_ins"]), int(r["n_del"])
    cols = c.nMatch + c.nMismatch + c.nIns + c.nDel
    c.pctSimilarity = (c.nMatch * 2.0) / (2 * cols) * 100.0 if cols else 0.0
    c.score, c.nCells = int(r["score"]), int(r["ncells"])


def PairwiseLocalAlign(seq_pairs, ks, params: MappingParameters, ctx, alignType=_lib.KB_FIT, qvs=None):
    """PairwiseLocalAlign(qSeq, tSeq, k, params, alignment, mappingBuffers, alignType)
    (/root/reference/iblasr/BlasrAlignImpl.hpp:1226-1309) for a list of (query, target) pairs in one call.
    Without usable quality values: AffineKBandAlign(qSeq, tSeq, SMRTDistanceMatrix, indel+2, indel-3,
    indel+2, indel-1, indel, k*1.2, ..., Global) (:1259-1267).  With them: KBandAlign(..., indel+2, indel+2,
    k, ..., idsScoreFn or scoreFn, alignType) (:1278-1294; the IDS scorer when the read has insertion
    QVs).  Then ComputeAlignmentStats with distScoreFn2 (:1304).  qvs: one QualityValues (or None) per
    pair.  Returns (pair results, blocks)."""
    n = len(seq_pairs)
    qvs = qvs if qvs is not None else [None] * n
    plain = [i for i in range(n) if not ReadHasMeaningfulQualityValues(qvs[i])]
    withqv = [i for i in range(n) if ReadHasMeaningfulQualityValues(qvs[i])]
    pairs_out = np.zeros(n, dtype=_lib.PAIR_RESULT_DT)
    blocks_out = [np.zeros(0, BLOCK_DT)] * n

    def run(idx, P, ks_, profile_parts=None):
        if not idx:
            return
        b = _kband_batch([seq_pairs[i] for i in idx], ks_, profile_parts)
        pr, bl, rc = ctx.kband_host(b, P)
        for x, i in enumerate(idx):
            pairs_out[i] = pr[x]
            blocks_out[i] = bl[int(pr[x]["blk_off"]):int(pr[x]["blk_off"]) + int(pr[x]["n_blk"])].copy() if int(pr[x]["status"]) == 0 else np.zeros(0, BLOCK_DT)

    run(plain, _kband_params(params, _lib.KB_GLOBAL, _lib.KB_AFFINE_HP, ins=params.indel, dele=params.indel),
        [int(ks[i] * 1.2) for i in plain])
    if withqv:
        Pq = _kband_params(params, alignType, ins=params.indel + 2, dele=params.indel + 2)
        run(withqv, Pq, [int(ks[i]) for i in withqv],
            [ids_profile(bytes(seq_pairs[i][0]), qvs[i], params) if qvs[i].insertionQV is not None
             else qv_profile(bytes(seq_pairs[i][0]), qvs[i], params) for i in withqv])
    return pairs_out, blocks_out


def refine_legacy(read: bytes, cands: List[AlignmentCandidate], params: MappingParameters, ctx, qv: Optional[QualityValues] = None):
    """The branch of RefineAlignment without guided alignment (-noUseGuidedAlign,
    /root/reference/iblasr/BlasrUtilsImpl.hpp:775-902): the two sequences are cut to the span of the SDP
    blocks, k = params.bandSize or |drift| * 1.5, at least 10 (:848-861), the positions move into the
    aligned-sequence offsets (:863-887), and PairwiseLocalAlign(qSeq, tSeq, k, ..., Fit) aligns them
    afresh (:901).  A candidate without blocks gets score 0 (:800-803)."""
    todo, seqs, ks = [], [], []
    for c in cands:
        if len(c.blocks) == 0:
            c.score = 0
            continue
        last = c.blocks[len(c.blocks) - 1]
        ql = int(last["q_pos"]) + int(last["length"])                     # qIsSubstring form (:826-830)
        tl = int(last["t_pos"]) + int(last["length"]) - int(c.blocks[0]["t_pos"])
        qs = c.qPos + c.qAlignedSeqPos
        k = params.bandSize if params.bandSize else int(abs(compute_drift(c.blocks)) * 1.5)
        k = max(k, 10)
        seqs.append((read[qs:qs + ql], c.tAlignedSeq[c.tPos:c.tPos + tl])); ks.append(k); todo.append(c)
        c.tAlignedSeqPos += c.tPos
        c.tAlignedSeqLength = tl
        c.qAlignedSeqPos += c.qPos
        c.qAlignedSeqLength = int(last["q_pos"]) + int(last["length"]) - int(c.blocks[0]["q_pos"])
        c.tPos = c.qPos = 0
        c.tAlignedSeq, c.qAlignedSeq = seqs[-1][1], seqs[-1][0]
    if not todo:
        return cands
    qvs = None
    if qv is not None:
        cut = lambda b, s, l: b[s:s + l] if b is not None else None
        qvs = []
        for c, (qs_, _) in zip(todo, seqs):
            s, l = c.qAlignedSeqPos, len(qs_)
            qvs.append(QualityValues(*[cut(getattr(qv, f), s, l) for f in
                                       ("qual", "insertionQV", "deletionQV", "deletionTag", "substitutionQV", "substitutionTag")]))
    pairs, blocks = PairwiseLocalAlign(seqs, ks, params, ctx, _lib.KB_FIT, qvs)
    for c, r, b in zip(todo, pairs, blocks):
        c.status = int(r["status"])
        c.blocks = b
        c.nBlocks = len(b)
        if c.status != _lib.PAIR_OK:
            continue
        c.qPos, c.tPos = int(r["q_pos"]), int(r["t_pos"])
        _stats_into(c, r)
        c.qInterval = (c.qAlignedSeqPos + int(r["q_start"]), c.qAlignedSeqPos + int(r["q_end"]))
        c.tInterval = (c.tAlignedSeqPos + int(r["t_start"]), c.tAlignedSeqPos + int(r["t_end"]))
    return cands


def refine_global_qv(read: bytes, cands: List[AlignmentCandidate], params: MappingParameters, ctx, qv: QualityValues,
                     subread_start: int = 0, subread_length: Optional[int] = None):
    """The -global branch (:648-675) as written, with the IDS scorer on the read's quality values:
    KBandAlign(subread, tAlignedSeq, SMRTDistanceMatrix, params.insertion, params.deletion, drift, ...,
    idsScoreFn, Global), RemoveEndGaps, ComputeAlignmentStats with distScoreFn2.  One call for all
    candidates (the band is per pair)."""
    if subread_length is None:
        subread_length = len(read) - subread_start
    sub = read[subread_start:subread_start + subread_length]
    cut = lambda b: b[subread_start:subread_start + subread_length] if b is not None else None
    prof = ids_profile(sub, QualityValues(*[cut(getattr(qv, f)) for f in
                                            ("qual", "insertionQV", "deletionQV", "deletionTag", "substitutionQV", "substitutionTag")]), params)
    if not cands:
        return cands
    seqs = [(sub, c.tAlignedSeq) for c in cands]
    ks = [compute_drift(c.blocks) for c in cands]
    b = _kband_batch(seqs, ks, [prof] * len(cands))
    pairs, blocks, rc = ctx.kband_host(b, _kband_params(params, _lib.KB_GLOBAL))
    for c, r in zip(cands, pairs):
        c.status = int(r["status"])
        if c.status != _lib.PAIR_OK:
            continue
        c.blocks = blocks[int(r["blk_off"]):int(r["blk_off"]) + int(r["n_blk"])].copy()
        c.tPos, c.qPos = int(r["t_pos"]), int(r["q_pos"]) + subread_start       # blocks already start at their first block
        c.score = int(r["score"])
    return cands


_COMP = bytes.maketrans(b"ACGTacgtNn", b"TGCAtgcaNn")


def _genome_slice(genome: bytes, pos: int, length: int, strand: int):
    """genome.Copy(pos, length) on the forward strand, CopyAsRC / MakeRC on the reverse strand, where
    positions count along the reverse complement (BlasrAlignImpl.hpp:1074-1079, :1135-1140)."""
    if strand == 0:
        return genome[pos:pos + length]
    g = len(genome)
    return genome[g - pos - length:g - pos].translate(_COMP)[::-1]


MAXIMUM_EXTEND_LENGTH = 500      # BlasrAlignImpl.hpp:1039


def ExtendAlignments(read: bytes, cands: List[AlignmentCandidate], genome: bytes, params: MappingParameters, ctx,
                     extendBandSize: int = 10, maxExtendDropoff: int = 10, contigStart: int = 0, contigEnd: Optional[int] = None):
    """The -extend step of AlignIntervals (/root/reference/iblasr/BlasrAlignImpl.hpp:1011-1201) for all
    candidates of a read in one call: behind the last block up to 500 bases of read and genome go
    through ExtendAlignmentForward (:1086-1099), in front of the first block up to 500 through
    ExtendAlignmentReverse (:1144-1155), both with distScoreFn, extendBandSize and maxExtendDropoff; an
    extension with a negative score is spliced on (:1102-1120, :1158-1196).  Positions of a
    reverse-strand candidate count along the reverse complement of the genome, as the reference's do.
    Counts and score of the pieces add up (ComputeAlignmentStats with distScoreFn2 at :1214 rescans
    the whole alignment; with its linear gap costs that is the sum).  Returns the number of ends extended."""
    contigEnd = len(genome) if contigEnd is None else contigEnd
    # The prelude of :1020-1037, for every candidate: the aligned sequences are cut to the alignment's own
    # boundaries (from the first block's start over QEnd() / TEnd() bases) and qPos / tPos become 0, so that
    # what is spliced on below lands directly behind the last aligned base and in front of the first.
    for c in cands:
        nb = len(c.blocks)
        qend = int(c.blocks[nb - 1]["q_pos"]) + int(c.blocks[nb - 1]["length"]) if nb else 0
        tend = int(c.blocks[nb - 1]["t_pos"]) + int(c.blocks[nb - 1]["length"]) if nb else 0
        c.qAlignedSeqPos += c.qPos
        c.qAlignedSeqLength = qend
        c.qAlignedSeq = read[c.qAlignedSeqPos:c.qAlignedSeqPos + qend]
        c.qPos = 0
        c.tAlignedSeqPos += c.tPos
        c.tAlignedSeq = c.tAlignedSeq[c.tPos:c.tPos + tend]
        c.tAlignedSeqLength = tend
        c.tPos = 0
    jobs, seqs, rev = [], [], []
    for ci, c in enumerate(cands):
        if len(c.blocks) == 0:
            continue
        last = c.blocks[len(c.blocks) - 1]
        lastQ = int(last["q_pos"]) + int(last["length"]) + c.qAlignedSeqPos
        lastT = int(last["t_pos"]) + int(last["length"]) + c.tAlignedSeqPos
        rl = min(len(read) - lastQ, MAXIMUM_EXTEND_LENGTH)
        gl = min(contigEnd - lastT, MAXIMUM_EXTEND_LENGTH)
        if rl > 0 and gl > 0:
            jobs.append((ci, 0, lastQ, lastT, rl, gl)); rev.append(0)
            seqs.append((read[lastQ:lastQ + rl], _genome_slice(genome, lastT, gl, c.tStrand)))
        firstQ, firstT = c.qAlignedSeqPos, c.tAlignedSeqPos
        rl = min(firstQ, MAXIMUM_EXTEND_LENGTH)
        gl = min(firstT - contigStart, MAXIMUM_EXTEND_LENGTH)
        if rl > 0 and gl > 0:
            jobs.append((ci, 1, firstQ - rl, firstT - gl, rl, gl)); rev.append(1)
            seqs.append((read[firstQ - rl:firstQ], _genome_slice(genome, firstT - gl, gl, c.tStrand)))
    if not jobs:
        return 0
    # AlignIntervals' own distScoreFn(SMRTDistanceMatrix, params.insertion, params.deletion) (BlasrAlignImpl.hpp:576), read as ctor(ins, del)
    P = _kband_params(params, _lib.KB_EXTEND, ins=params.insertion, dele=params.deletion, max_drops=maxExtendDropoff)
    pairs, blocks, rc = ctx.kband_host(_kband_batch(seqs, [extendBandSize] * len(jobs), reverse=rev), P)
    done = 0
    # forward extensions first, then the reverse ones, as the reference does per candidate
    for order in (0, 1):
        for (ci, side, q0, t0, rl, gl), r, (qs, ts) in zip(jobs, pairs, seqs):
            if side != order or int(r["status"]) != _lib.PAIR_OK or int(r["dp_score"]) >= 0 or int(r["n_blk"]) == 0:
                continue
            c = cands[ci]
            eb = blocks[int(r["blk_off"]):int(r["blk_off"]) + int(r["n_blk"])]
            eq, et = int(r["q_pos"]), int(r["t_pos"])
            qend, tend = eq + int(eb[-1]["q_pos"]) + int(eb[-1]["length"]), et + int(eb[-1]["t_pos"]) + int(eb[-1]["length"])
            oq, ot = c.qPos + c.qAlignedSeqPos, c.tPos + c.tAlignedSeqPos      # origin of the candidate's blocks
            if side == 0:
                # :1111-1119: the genome suffix up to the extension's end goes behind tAlignedSeq, the query grows by as much
                nb = eb.copy()
                nb["q_pos"] += np.uint32(q0 + eq - oq); nb["t_pos"] += np.uint32(t0 + et - ot)
                c.blocks = np.concatenate([c.blocks, nb])
                c.tAlignedSeq = c.tAlignedSeq + ts[:tend]
                c.qAlignedSeqLength = len(c.qAlignedSeq) + qend
                c.qAlignedSeq = read[c.qAlignedSeqPos:c.qAlignedSeqPos + c.qAlignedSeqLength]
                c.tAlignedSeqLength = len(c.tAlignedSeq)
            else:
                # :1166-1183: the WHOLE genome prefix goes in front (genomePrefix.Append(tAlignedSeq, length - tPos) with
                # tPos = 0 after the prelude), the extension's own tPos / qPos say where in it the alignment starts
                ob = c.blocks.copy()
                ob["q_pos"] += np.uint32(oq - (q0 + eq)); ob["t_pos"] += np.uint32(ot - (t0 + et))
                c.blocks = np.concatenate([eb, ob])
                c.tAlignedSeq = ts + c.tAlignedSeq
                old_qlen = len(c.qAlignedSeq)
                c.tAlignedSeqPos, c.qAlignedSeqPos = t0, q0
                c.tPos, c.qPos = et, eq
                c.qAlignedSeqLength = rl + old_qlen
                c.qAlignedSeq = read[q0:q0 + c.qAlignedSeqLength]
                c.tAlignedSeqLength = len(c.tAlignedSeq)
            c.nBlocks = len(c.blocks)
            c.nMatch += int(r["n_match"]); c.nMismatch += int(r["n_mismatch"])
            c.nIns += int(r["n_ins"]); c.nDel += int(r["n_del"])
            c.score += int(r["score"])
            cols = c.nMatch + c.nMismatch + c.nIns + c.nDel
            c.pctSimilarity = (c.nMatch * 2.0) / (2 * cols) * 100.0 if cols else 0.0
            first, lastb = c.blocks[0], c.blocks[-1]
            base_q, base_t = c.qPos + c.qAlignedSeqPos, c.tPos + c.tAlignedSeqPos
            c.qInterval = (base_q + int(first["q_pos"]), base_q + int(lastb["q_pos"]) + int(lastb["length"]))
            c.tInterval = (base_t + int(first["t_pos"]), base_t + int(lastb["t_pos"]) + int(lastb["length"]))
            done += 1
    return done
