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
    maxBandSize: int = 250   # libcpp's MAX_BAND_SIZE (ctest/concordant.t:23 mentions it)
    tieBreak: int = _lib.TIE_DIAG_LEFT_UP
    sdpTupleSize: int = 11   # :222
    sdpIns: int = 5          # :220
    sdpDel: int = 10         # :221
    sdpMaxFragments: int = 1024   # not a BLASR option: the device's budget of maximal matches per candidate
    sdpBypassThreshold: float = 1000000.0   # :340 (0.75 in the -sdpbypass preset, :721)
    scoreMatrix: Optional[list] = None     # 5x5, -scoreMatrix (Blasr.cpp:922-937)


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


def scoring_from_params(p: MappingParameters):
    """distScoreFn(SMRTDistanceMatrix, params.deletion, params.insertion) read as ctor(ins, del)
    (:630-631) and distScoreFn2(indel, indel) (:633-634); band per :706-731.  The affine costs of
    both scorers are the constructor defaults above, NOT params.affineOpen / affineExtend."""
    if p.doGlobalAlignment:
        raise ValueError("the -global branch has its own scorer and band: use scoring_global() / refine_global()")
    mode = _lib.MODE_AFFINE if p.affineAlign else _lib.MODE_LINEAR
    band = p.bandSize if p.affineAlign else p.guidedAlignBandSize
    return make_scoring(match=p.match, mismatch=p.mismatch, ins=p.deletion, dele=p.insertion,
                        indel=p.indel, band=band, max_band=p.maxBandSize, mode=mode,
                        tie_break=p.tieBreak, affine_open=DIST_SCORER_AFFINE_OPEN, affine_extend=DIST_SCORER_AFFINE_EXTEND,
                        stats_affine=1 if p.affineAlign else 0, stats_affine_open=STATS_SCORER_AFFINE_OPEN,
                        stats_affine_extend=STATS_SCORER_AFFINE_EXTEND, sub=p.scoreMatrix)


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
    sends every candidate to SDPAlign.  The -rbao variant (SDP between consecutive anchors only,
    :785-978) is not mirrored."""
    n_sub = subreadLength if subreadLength is not None else len(read)
    todo = []
    for c, a in zip(cands, anchors):
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


def GuidedAlignFromSequences(read: bytes, cands: List[AlignmentCandidate], params: MappingParameters, ctx,
                             bandSize: int = 12, sdpTupleSize: int = 6):
    """The GuidedAlign overload that takes only the two sequences: it chains matches of sdpTupleSize
    bases itself and runs the banded DP around that chain.  AlignSubreadToAlignmentTarget uses it to
    align every subread of a ZMW to the window its CCS (or longest subread) mapped to:
    GuidedAlign(subread, alignedRefSequence, idsScoreFn, 12, params.sdpIns, params.sdpDel,
    params.indelRate, mappingBuffers, exploded, Local, computeProbIsFalse, 6)
    (/root/reference/iblasr/BlasrAlignImpl.hpp:1418-1421), then ComputeAlignmentStats with
    distScoreFn2 (:1448-1450).  Here: SDPAlignCandidates with the short tuple, then the refine stage
    with band `bandSize`, all candidates of the read in one batch each.  What differs from that call
    site: the DP runs on the distance matrix with params.insertion / params.deletion, not on the
    QV-aware IDS scorer (its formulas are libcpp's and need per-base quality values), and it is
    anchored at the chain's first and last block (the overload's Local end handling is libcpp's)."""
    import dataclasses
    p = dataclasses.replace(params, sdpTupleSize=sdpTupleSize, guidedAlignBandSize=bandSize, useGuidedAlign=True,
                            doGlobalAlignment=False, refineAlignments=True)
    SDPAlignCandidates(read, cands, p, ctx)
    return RefineAlignments(read, cands, p, ctx)


def RefineAlignments(read: bytes, cands: List[AlignmentCandidate], params: MappingParameters, ctx, want_blocks: bool = True):
    """Batched RefineAlignments (:905-923): all candidates of the read in one submit.
    want_blocks False: for interval / summary output (printFormat Interval, SummaryPrint, :992-1000), which
    prints only what the pair results hold: no block leaves the device."""
    if not params.refineAlignments or not cands:
        return cands
    if params.doGlobalAlignment:
        refine_global(read, cands, params, ctx)
        if params.sortRefinedAlignments:
            cands.sort(key=lambda c: c.score)
        return cands
    if not params.useGuidedAlign:
        raise NotImplementedError("the legacy -noUseGuidedAlign branch (:775-902) is not on the accelerated path")
    ctx.set_scoring(scoring_from_params(params))
    batch = build_batch(read, cands)
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
    """Drop from the first candidate that has no blocks or scores above maxScore
    (iblasr/BlasrUtilsImpl.hpp:485-509), as MapRead does right after refining."""
    for i, c in enumerate(cands):
        if c.n_blocks() == 0 or c.score > params.maxScore:
            del cands[i:]
            break
    return cands
