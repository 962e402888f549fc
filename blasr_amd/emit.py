"""Text records from refined candidates: CIGAR operations and the m4 line.

The reference prints a refined candidate through libcpp's writers
(/root/reference/iblasr/BlasrUtilsImpl.hpp:950-1008: SAMOutput / BAMOutput::PrintAlignment with
params.cigarUseSeqMatch and params.allowAdjacentIndels, IntervalOutput::Print for -m 4).  Those
writers are not part of the reference tree, so what is restated here is what the reference's own
files show of their output:

* CIGAR alphabet and shape: runs of `=`, `X`, `I`, `D` (and `S` clips) as in the two literal
  strings of ctest/bamOut.t:21-22 (`--bam` always sets cigarUseSeqMatch, MappingParameters.h:589);
  with cigarUseSeqMatch off a gap-free run is one `M`.
* Adjacent insertions and deletions "are merged into one operation" unless -allowAdjacentIndels
  (RegisterBlasrOptions.h:254-259; ctest/cigarAdjecentIndels.t:12-16 greps for `ID` / `DI`): the
  part the two gaps have in common is emitted as aligned columns (compared base by base under
  cigarUseSeqMatch), the rest as one insertion or one deletion.
* m4 columns: qName tName score pctSimilarity qStrand qStart qEnd qLength tStrand tStart tEnd
  tLength mapQV (ctest/filtercriteria.t:20 tests column 4 as the similarity, bug25328.t:12
  columns 6-7 as the query interval, hitpolicy.t:95 column 10 as the target start).

Parity with the writers themselves is unpinned (their sources and the test data are absent); the
tests check these functions against the kernels' own statistics and against the shape of the
reference's literals.
"""
from typing import List, Tuple

import numpy as np


def _match_ops(q: np.ndarray, t: np.ndarray, use_seq_match: bool) -> List[Tuple[int, str]]:
    """Aligned columns q[k] | t[k] as `M`, or as runs of `=` / `X`."""
    n = len(q)
    if n == 0:
        return []
    if not use_seq_match:
        return [(n, "M")]
    same = q == t
    edges = np.flatnonzero(same[1:] != same[:-1]) + 1
    starts = np.concatenate(([0], edges))
    ends = np.concatenate((edges, [n]))
    return [(int(e - s), "=" if same[s] else "X") for s, e in zip(starts, ends)]


def _push(ops: List[Tuple[int, str]], new: List[Tuple[int, str]]):
    for n, op in new:
        if n <= 0:
            continue
        if ops and ops[-1][1] == op:
            ops[-1] = (ops[-1][0] + n, op)
        else:
            ops.append((n, op))


def cigar_ops(blocks, q_seq: bytes, t_seq: bytes, q_pos: int = 0, t_pos: int = 0,
              use_seq_match: bool = True, allow_adjacent_indels: bool = False,
              clip_left: int = 0, clip_right: int = 0) -> List[Tuple[int, str]]:
    """CIGAR of a refined candidate as (count, op) pairs.

    blocks: gap-free runs (q_pos, t_pos, length) relative to (q_pos, t_pos) inside q_seq / t_seq,
    i.e. candidate.blocks with candidate.qPos / tPos and qAlignedSeq / tAlignedSeq after
    RefineAlignment's copy-back (BlasrUtilsImpl.hpp:745-772).  The alignment runs from the first
    block to the end of the last one; clip_left / clip_right add `S` operations for the read bases
    outside it (--clipping soft)."""
    q = np.frombuffer(q_seq, dtype=np.uint8)
    t = np.frombuffer(t_seq, dtype=np.uint8)
    ops: List[Tuple[int, str]] = []
    _push(ops, [(clip_left, "S")])
    nb = len(blocks)
    for b in range(nb):
        bq, bt, bl = int(blocks[b]["q_pos"]) + q_pos, int(blocks[b]["t_pos"]) + t_pos, int(blocks[b]["length"])
        _push(ops, _match_ops(q[bq:bq + bl], t[bt:bt + bl], use_seq_match))
        if b + 1 == nb:
            break
        nq, nt = int(blocks[b + 1]["q_pos"]) + q_pos, int(blocks[b + 1]["t_pos"]) + t_pos
        q_gap, t_gap = nq - (bq + bl), nt - (bt + bl)
        if q_gap < 0 or t_gap < 0:
            raise ValueError("blocks overlap or are not increasing at block %d" % b)
        if allow_adjacent_indels or q_gap == 0 or t_gap == 0:
            _push(ops, [(q_gap, "I"), (t_gap, "D")])
        else:
            common = min(q_gap, t_gap)
            _push(ops, _match_ops(q[bq + bl:bq + bl + common], t[bt + bl:bt + bl + common], use_seq_match))
            _push(ops, [(q_gap - common, "I"), (t_gap - common, "D")])
    _push(ops, [(clip_right, "S")])
    return ops


def cigar(blocks, q_seq: bytes, t_seq: bytes, **kw) -> str:
    return "".join("%d%s" % (n, op) for n, op in cigar_ops(blocks, q_seq, t_seq, **kw))


def aligned_interval(blocks, q_pos: int = 0, t_pos: int = 0):
    """(qStart, qEnd, tStart, tEnd) of the alignment inside the aligned sequences, end exclusive."""
    if len(blocks) == 0:
        return q_pos, q_pos, t_pos, t_pos
    first, last = blocks[0], blocks[len(blocks) - 1]
    return (q_pos + int(first["q_pos"]), q_pos + int(last["q_pos"]) + int(last["length"]),
            t_pos + int(first["t_pos"]), t_pos + int(last["t_pos"]) + int(last["length"]))


def m4_line(cand, q_name: str, t_name: str, q_length: int, t_length: int, map_qv: int = 254,
            q_strand: int = 0) -> str:
    """One -m 4 record of a refined candidate (refine.AlignmentCandidate after RefineAlignments).
    Coordinates are those of the aligned strands; the caller passes names, full lengths and the
    mapping quality, which are computed outside the refine stage.  A candidate refined without a
    block arena (RefineAlignments(..., want_blocks=False)) carries its intervals from the pair result."""
    if len(cand.blocks) == 0 and getattr(cand, "qInterval", None) is not None:
        (qs, qe), (ts, te) = cand.qInterval, cand.tInterval
    else:
        qs, qe, ts, te = aligned_interval(cand.blocks, cand.qAlignedSeqPos + cand.qPos, cand.tAlignedSeqPos + cand.tPos)
    return "%s %s %d %g %d %d %d %d %d %d %d %d %d" % (
        q_name, t_name, cand.score, cand.pctSimilarity, q_strand, qs, qe, q_length,
        cand.tStrand, ts, te, t_length, map_qv)


# ---- the record formats with the aligned strings: -m 5 (CompareSequencesParsable) and SAM

_RC = bytes.maketrans(b"ACGTacgtNn", b"TGCAtgcaNn")


def aligned_strings(blocks, q_seq: bytes, t_seq: bytes, q_pos: int = 0, t_pos: int = 0):
    """(qAlignedSeq, matchPattern, tAlignedSeq) of a candidate from its first block to the end of its last:
    '-' against a gap, query-only stretches of a gap before its target-only ones, '|' for equal bases
    (case-insensitive) and '*' otherwise in the pattern, as `blasr -m 5` prints them."""
    q, t = q_seq.decode("latin1"), t_seq.decode("latin1")
    qs, ts = [], []
    for b in range(len(blocks)):
        bq, bt, ln = int(blocks[b]["q_pos"]) + q_pos, int(blocks[b]["t_pos"]) + t_pos, int(blocks[b]["length"])
        qs.append(q[bq:bq + ln]); ts.append(t[bt:bt + ln])
        if b + 1 < len(blocks):
            nq, nt = int(blocks[b + 1]["q_pos"]) + q_pos, int(blocks[b + 1]["t_pos"]) + t_pos
            qs.append(q[bq + ln:nq] + "-" * (nt - bt - ln)); ts.append("-" * (nq - bq - ln) + t[bt + ln:nt])
    qa, ta = "".join(qs), "".join(ts)
    pat = "".join("|" if a == b and a != "-" else "*" for a, b in zip(qa.upper(), ta.upper()))
    return qa, pat, ta


def m5_line(cand, q_name: str, t_name: str, q_length: int, t_length: int, map_qv: int = 254, q_strand: int = 0) -> str:
    """One -m 5 record (printFormat CompareSequencesParsable, iblasr/RegisterBlasrOptions.h:276) of a refined
    candidate that holds its blocks and aligned sequences: qName qLength qStart qEnd qStrand tName tLength tStart tEnd
    tStrand score numMatch numMismatch numIns numDel mapQV qAlignedSeq matchPattern tAlignedSeq (layout from BLASR's
    manual; the writer itself is libcpp code).  Coordinates as in m4_line; strands print as + / -."""
    qs, qe, ts, te = aligned_interval(cand.blocks, cand.qAlignedSeqPos + cand.qPos, cand.tAlignedSeqPos + cand.tPos)
    qa, pat, ta = aligned_strings(cand.blocks, cand.qAlignedSeq, cand.tAlignedSeq, cand.qPos, cand.tPos)
    return " ".join(str(x) for x in (q_name, q_length, qs, qe, "-" if q_strand else "+", t_name, t_length, ts, te,
                                     "-" if cand.tStrand else "+", cand.score, cand.nMatch, cand.nMismatch, cand.nIns, cand.nDel,
                                     map_qv, qa, pat, ta))


def sam_line(cand, q_name: str, t_name: str, read: bytes, t_length: int, map_qv: int = 254, use_seq_match: bool = False,
             allow_adjacent_indels: bool = False, soft_clip: bool = True, edit_distance=None) -> str:
    """One SAM record of a refined candidate (what SAMOutput::PrintAlignment makes of it, iblasr/BlasrUtilsImpl.hpp:969-971;
    the writer is libcpp code: the eleven mandatory columns and AS / NM follow the SAM specification).  The read is the
    forward read; a candidate on the reverse strand (tStrand 1: its window is the reverse complement of the reference)
    is reported on the forward strand of the reference: flag 16, the read reverse-complemented, the CIGAR reversed, POS
    counted from the window's far end.  t_length: length of the reference sequence the positions of a reverse-strand
    candidate count along.  Unclipped read bases become soft clips (--clipping soft) or are left out (none)."""
    qs, qe, ts, te = aligned_interval(cand.blocks, cand.qAlignedSeqPos + cand.qPos, cand.tAlignedSeqPos + cand.tPos)
    ops = cigar_ops(cand.blocks, cand.qAlignedSeq, cand.tAlignedSeq, q_pos=cand.qPos, t_pos=cand.tPos,
                    use_seq_match=use_seq_match, allow_adjacent_indels=allow_adjacent_indels,
                    clip_left=qs if soft_clip else 0, clip_right=len(read) - qe if soft_clip else 0)
    seq = read if soft_clip else read[qs:qe]
    flag, pos = 0, ts + 1
    if cand.tStrand:
        flag, pos = 16, t_length - te + 1
        ops = ops[::-1]
        seq = seq.translate(_RC)[::-1]
    tags = ["AS:i:%d" % cand.score]
    if edit_distance is not None:
        tags.append("NM:i:%d" % edit_distance)
    return "\t".join([q_name, str(flag), t_name, str(pos), str(map_qv), "".join("%d%s" % o for o in ops), "*", "0", "0",
                      seq.decode("latin1"), "*"] + tags)
