"""Mapping qualities of a read's refined candidates: host-side mirror of StoreMapQVs
(/root/reference/iblasr/BlasrUtilsImpl.hpp:108-309) and its helpers AlignmentsOverlap (:369-409),
PartitionOverlappingAlignments (:411-444), SumMismatches (:344-366, the branch without quality
values) and ScaleMapQVByClusterSize (:97-106).

What the device contributes is each candidate's path rescored with the probability scorer
(SMRTLogProbMatrix + insertion / deletion / affine costs, :117-123, :145-149): one of the extra
rescoring scorers of gba_scoring, delivered in gba_pair_result.extra_score.  Everything else here
is arithmetic over a handful of candidates per read.

Three small functions of libcpp are restated because it is not in the reference tree (UNPINNED):
GetQInterval (the aligned stretch of the read, forward strand), LogSumOfTwo (log10 of a sum of two
powers of ten) and Phred (-10 log10, truncated to an integer).
"""
import math
from typing import List, Sequence

import numpy as np

# probScore is a float field of the candidate and the denominator a `float mapQVDenominator` (:238): sums and
# differences are rounded to single precision where the reference stores or subtracts them, the exponentials run
# in double (`double sub ... exp(log10*sub)`, :277-280).  Phred() truncates, so the rounding can move a boundary case by one.
_f32 = np.float32

MAX_PHRED_SCORE = 254            # iblasr/BlasrHeaders.h:19
MAPQV_END_ALIGN_WIGGLE = 5       # iblasr/BlasrHeaders.h:20


def q_interval(c):
    """GetQInterval: [start, end) of the read the candidate aligns (all reads are forward).  A
    candidate refined without a block arena carries the interval itself (from the pair result)."""
    if len(c.blocks) == 0 and getattr(c, "qInterval", None) is not None:
        return c.qInterval
    if len(c.blocks) == 0:
        s = c.qAlignedSeqPos + c.qPos
        return s, s
    last = c.blocks[len(c.blocks) - 1]
    s = c.qAlignedSeqPos + c.qPos
    return s + int(c.blocks[0]["q_pos"]), s + int(last["q_pos"]) + int(last["length"])


def log_sum_of_two(a: float, b: float) -> float:
    """log10(10^a + 10^b), computed from the larger exponent."""
    hi, lo = (a, b) if a > b else (b, a)
    return hi + math.log10(1.0 + 10.0 ** (lo - hi))


def phred(p: float) -> int:
    return int(-10.0 * math.log10(p))


def alignments_overlap(a, b, min_percent_overlap: float) -> bool:
    """:369-409."""
    a0, a1 = q_interval(a)
    b0, b1 = q_interval(b)
    ovp = 0
    if a0 <= b0 and a1 >= b1:
        return True
    if b0 <= a0 and b1 >= a1:
        return True
    if b0 <= a1 <= b1:
        ovp = a1 - b0
    elif b0 <= a0 <= b1:
        ovp = b1 - a0
    pct = 0.0
    if a1 - a0 > 0 and b1 - b0 > 0:
        pct = max(ovp / float(a1 - a0), ovp / float(b1 - b0))
    return pct > min_percent_overlap


def partition_overlapping_alignments(cands: Sequence, min_overlap: float) -> List[List[int]]:
    """:411-444: a candidate joins the first partition that holds one it overlaps (or spans)."""
    parts: List[List[int]] = []
    for i, c in enumerate(cands):
        found = False
        for part in parts:
            for j in part:
                ci0, ci1 = q_interval(c)
                cj0, cj1 = q_interval(cands[j])
                if alignments_overlap(c, cands[j], min_overlap) or (ci0 <= cj0 and ci1 > cj1):
                    part.append(i)          # a std::set in the reference: iteration below is by index
                    found = True
                    break
            if found:
                break
        if not found:
            parts.append([i])
    return [sorted(p) for p in parts]


def sum_mismatches(c, mismatch_score: int, full_start: int, full_end: int) -> int:
    """:344-366 without quality values: a flat price for every base of the partition's widest
    stretch that this candidate leaves unaligned."""
    s, e = q_interval(c)
    return mismatch_score * ((s - full_start) + (full_end - e))


def store_map_qvs(cands: Sequence, raw_prob_scores: Sequence[int], min_fraction_overlapping: float = 0.75,
                  n_candidates: int = 10, scale_by_clusters: bool = False):
    """StoreMapQVs: sets c.probScore and c.mapQV of every candidate.  raw_prob_scores[i] is the
    candidate's path under the probability scorer (ComputeAlignmentScore, what the device returns);
    probScore = -raw / 10 (:145-149).  min_fraction_overlapping is
    params.minFractionToBeConsideredOverlapping, n_candidates params.nCandidates."""
    for c, raw in zip(cands, raw_prob_scores):
        c.probScore = float(_f32(-float(raw) / 10.0))
    parts = partition_overlapping_alignments(cands, min_fraction_overlapping)
    for part in parts:
        # the widest aligned stretch of the partition (:170-191)
        begin, end = q_interval(cands[part[0]])
        for i in part[1:]:
            s, e = q_interval(cands[i])
            if e - s > end - begin:
                begin, end = s, e
        # candidates that stop short of it pay for the missing ends (:214-233)
        for i in part:
            s, e = q_interval(cands[i])
            mm = 0
            if s - begin > MAPQV_END_ALIGN_WIGGLE or end - e > MAPQV_END_ALIGN_WIGGLE:
                mm = sum_mismatches(cands[i], 15, begin, end)
            cands[i].probScore = float(_f32(cands[i].probScore + -mm * 0.5))
        denom = cands[part[0]].probScore
        for i in part[1:]:
            denom = float(_f32(log_sum_of_two(denom, cands[i].probScore)))
        for i in part:
            c = cands[i]
            sub = float(_f32(c.probScore) - _f32(denom))       # float - float
            if len(part) == 1:
                c.mapQV = MAX_PHRED_SCORE                       # :266-268
            elif sub < -20:
                c.mapQV = 0                                     # :273-275
            else:
                expo = math.exp(math.log(10) * sub)
                diff = 1.0 - expo
                if expo == 0:
                    q = 0
                elif diff == 0:
                    q = MAX_PHRED_SCORE
                else:
                    q = phred(diff)
                c.mapQV = min(q, MAX_PHRED_SCORE)
            if scale_by_clusters:                               # ScaleMapQVByClusterSize, :97-106
                nsc = getattr(c, "numSignificantClusters", 1)
                if nsc > n_candidates:
                    c.mapQV = phred((1 - 10.0 ** (c.mapQV / -10.0)) * (float(n_candidates) / nsc))
                elif nsc == 0:
                    c.mapQV = 0
    return cands
