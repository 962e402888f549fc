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
from .api import make_scoring

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


def scoring_from_params(p: MappingParameters):
    """distScoreFn(SMRTDistanceMatrix, params.deletion, params.insertion) read as ctor(ins, del)
    (:630-631) and distScoreFn2(indel, indel) (:633-634); band per :706-731."""
    if p.doGlobalAlignment:
        raise ValueError("use refine_global() for -global")
    mode = _lib.MODE_AFFINE if p.affineAlign else _lib.MODE_LINEAR
    band = p.bandSize if p.affineAlign else p.guidedAlignBandSize
    return make_scoring(match=p.match, mismatch=p.mismatch, ins=p.deletion, dele=p.insertion,
                        indel=p.indel, band=band, max_band=p.maxBandSize, mode=mode,
                        tie_break=p.tieBreak, affine_open=p.affineOpen, affine_extend=p.affineExtend,
                        stats_affine=1 if p.affineAlign else 0, sub=p.scoreMatrix)


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
    """:733-772 for every candidate the kernels aligned."""
    for i, (c, r) in enumerate(zip(cands, pairs)):
        c.status = int(r["status"])
        if c.status != _lib.PAIR_OK:
            continue
        c.blocks = blocks[int(r["blk_off"]):int(r["blk_off"]) + int(r["n_blk"])].copy()
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


def RefineAlignments(read: bytes, cands: List[AlignmentCandidate], params: MappingParameters, ctx):
    """Batched RefineAlignments (:905-923): all candidates of the read in one submit."""
    if not params.refineAlignments or not cands:
        return cands
    if not params.useGuidedAlign:
        raise NotImplementedError("the legacy -noUseGuidedAlign branch (:775-902) is not on the accelerated path")
    ctx.set_scoring(scoring_from_params(params))
    batch = build_batch(read, cands)
    pairs, blocks, rc = ctx.refine_host(batch)
    copy_back(read, cands, batch, pairs, blocks)
    if params.sortRefinedAlignments:
        cands.sort(key=lambda c: c.score)        # stable, ascending: lower is better (:920-922)
    return cands


def RemoveLowQualityAlignments(cands: List[AlignmentCandidate], params: MappingParameters):
    """Drop from the first candidate that has no blocks or scores above maxScore
    (iblasr/BlasrUtilsImpl.hpp:485-509), as MapRead does right after refining."""
    for i, c in enumerate(cands):
        if len(c.blocks) == 0 or c.score > params.maxScore:
            del cands[i:]
            break
    return cands
