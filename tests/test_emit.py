"""CIGAR / m4 text from refined candidates (blasr_amd/emit.py).

The reference's writers are not in its tree, so these tests tie the text to what IS pinned here:
the statistics the oracle (and, bit-identically, the kernels) compute for the same path, and the
shape of the literal CIGARs in the reference's ctest/bamOut.t:21-22."""
import re

import numpy as np
import pytest

from blasr_amd import emit, refine
from blasr_amd.synth import BLOCK_DT

# the two known answers the reference's tests hold (ctest/bamOut.t:21-22); their inputs are absent,
# so they pin the alphabet and the run structure only
REF_CIGARS = (
    "25=1I28=1I41=1I5=1D6=1X12=1I15=1I2=1I16=1D10=1I11=1I74=1D12=1D7=3I4=1I6=1D1=2D14=1D16=1I8=1D4=1D5=1D20=1I3=1I10=1I37=1I13=1I25=1I15=1I7=1I11=1I3=2I1=1I16=1I6=1I8=1I11=1X1=1I5=1I56=1I17=",
    "28=1D7=1I1=1I9=2I12=1I3=1D13=1I15=1I2=1X49=1I19=1I14=1I5=1D17=1D20=1D86=1I21=1I9=1I24=1I6=1I1=1I2=1D11=1D4=1D3=1D31=1D6=1I6=1I9=1I57=2I24=1I26=1I8=1I43=1S",
)
CIGAR_RE = re.compile(r"^(\d+[=XIDSM])+$")


def _blocks(lst):
    a = np.zeros(len(lst), dtype=BLOCK_DT)
    for i, b in enumerate(lst):
        a[i] = b
    return a


def _runs(s):
    return [(int(n), op) for n, op in re.findall(r"(\d+)([=XIDSM])", s)]


def test_reference_literals_have_the_shape_emitted_here():
    for s in REF_CIGARS:
        assert CIGAR_RE.match(s)
        ops = [op for _, op in _runs(s)]
        assert all(a != b for a, b in zip(ops, ops[1:]))            # runs are maximal
        assert "ID" not in re.sub(r"\d", "", s) and "DI" not in re.sub(r"\d", "", s)


def test_hand_cases():
    q = b"ACGTACGTAC"
    t = b"ACGTTCGTAC"
    assert emit.cigar(_blocks([(0, 0, 10)]), q, t) == "4=1X5="
    assert emit.cigar(_blocks([(0, 0, 10)]), q, t, use_seq_match=False) == "10M"
    # insertion of two query bases, then a deletion of one target base
    q2, t2 = b"AAAACCGGGGTTTT", b"AAAAGGGGATTTT"
    blk = _blocks([(0, 0, 4), (6, 4, 4), (10, 9, 4)])
    assert emit.cigar(blk, q2, t2) == "4=2I4=1D4="
    assert emit.cigar(blk, q2, t2, clip_left=3, clip_right=1) == "3S4=2I4=1D4=1S"
    # adjacent gaps: 2 query bases against 1 target base between the blocks
    q3, t3 = b"AAAACTGGGG", b"AAAAGGGGG"
    blk3 = _blocks([(0, 0, 4), (6, 5, 4)])
    assert emit.cigar(blk3, q3, t3, allow_adjacent_indels=True) == "4=2I1D4="
    assert emit.cigar(blk3, q3, t3) == "4=1X1I4="                   # C|G as a column, T inserted
    assert emit.cigar(blk3, q3, t3, use_seq_match=False) == "5M1I4M"
    # offsets of the aligned region inside the sequences
    assert emit.cigar(_blocks([(0, 0, 4)]), b"NN" + q, b"N" + t, q_pos=2, t_pos=1) == "4="
    with pytest.raises(ValueError):
        emit.cigar(_blocks([(0, 0, 4), (3, 5, 2)]), q, t)


def _consumed(ops):
    qn = sum(n for n, op in ops if op in "=XMIS")
    tn = sum(n for n, op in ops if op in "=XMD")
    return qn, tn


def test_cigar_agrees_with_the_path_statistics(oracle, synth):
    """Along the oracle's refined paths: `=` / `X` / `I` / `D` totals are nMatch / nMismatch /
    nIns / nDel of ComputeAlignmentStats once the gaps in front of the first block and behind the
    last one (outside the CIGAR: they become clips) are added back."""
    batch = synth.generate(24, read_len=1500, seed=77)
    # a mismatch dearer than an insertion plus a deletion, so that paths do hold adjacent gaps
    res, blks = oracle.align_batch(batch, oracle.make_params(band=10, mismatch=12))
    seen_adjacent = 0
    for p in range(24):
        r, b = res[p], blks[p]
        assert int(r["status"]) == 0
        q, t, _ = synth.pair(batch, p)
        q, t = bytes(q), bytes(t)
        qp, tp = int(r["q_pos"]), int(r["t_pos"])
        ops = emit.cigar_ops(b, q, t, q_pos=qp, t_pos=tp, allow_adjacent_indels=True)
        s = "".join("%d%s" % o for o in ops)
        assert CIGAR_RE.match(s)
        qs, qe, ts, te = emit.aligned_interval(b, qp, tp)
        assert _consumed(ops) == (qe - qs, te - ts)
        lead_q, lead_t = qs - qp, ts - tp
        tail_q, tail_t = qp + int(r["rows"]) - qe, tp + int(r["cols"]) - te
        tot = {k: sum(n for n, op in ops if op == k) for k in "=XID"}
        assert tot["="] == int(r["n_match"]) and tot["X"] == int(r["n_mismatch"])
        assert tot["I"] + lead_q + tail_q == int(r["n_ins"])
        assert tot["D"] + lead_t + tail_t == int(r["n_del"])
        # merged form: same consumption, no adjacent I/D, and only the merged columns differ
        mops = emit.cigar_ops(b, q, t, q_pos=qp, t_pos=tp)
        ms = re.sub(r"\d", "", "".join("%d%s" % o for o in mops))
        assert "ID" not in ms and "DI" not in ms
        assert _consumed(mops) == (qe - qs, te - ts)
        mtot = {k: sum(n for n, op in mops if op == k) for k in "=XID"}
        merged = tot["I"] - mtot["I"]
        assert merged == tot["D"] - mtot["D"] >= 0
        assert mtot["="] + mtot["X"] == tot["="] + tot["X"] + merged
        seen_adjacent += merged
        # M form: one op per aligned stretch
        plain = emit.cigar_ops(b, q, t, q_pos=qp, t_pos=tp, use_seq_match=False)
        assert sum(n for n, op in plain if op == "M") == mtot["="] + mtot["X"]
        # the column-by-column restatement under oracle/ gives the same operations in all four forms
        for sm in (True, False):
            for adj in (True, False):
                assert oracle.cigar_ops(b, q, t, qp, tp, use_seq_match=sm, allow_adjacent_indels=adj) == \
                    emit.cigar_ops(b, q, t, q_pos=qp, t_pos=tp, use_seq_match=sm, allow_adjacent_indels=adj), (p, sm, adj)
    assert seen_adjacent > 0        # the workload does exercise the merge


def test_cigar_oracle_hand_cases(oracle):
    """oracle/cigar_oracle.c on cases small enough to check by eye (the same strings test_hand_cases expects of emit)."""
    text = lambda ops: "".join("%d%s" % o for o in ops)
    q, t = b"ACGTACGTAC", b"ACGTTCGTAC"
    assert text(oracle.cigar_ops([(0, 0, 10)], q, t)) == "4=1X5="
    assert text(oracle.cigar_ops([(0, 0, 10)], q, t, use_seq_match=False)) == "10M"
    q2, t2 = b"AAAACCGGGGTTTT", b"AAAAGGGGATTTT"
    assert text(oracle.cigar_ops([(0, 0, 4), (6, 4, 4), (10, 9, 4)], q2, t2)) == "4=2I4=1D4="
    q3, t3 = b"AAAACTGGGG", b"AAAAGGGGG"
    blk3 = [(0, 0, 4), (6, 5, 4)]
    assert text(oracle.cigar_ops(blk3, q3, t3, allow_adjacent_indels=True)) == "4=2I1D4="
    assert text(oracle.cigar_ops(blk3, q3, t3)) == "4=1X1I4="
    assert text(oracle.cigar_ops(blk3, q3, t3, use_seq_match=False)) == "5M1I4M"
    assert text(oracle.cigar_ops([(0, 0, 4)], b"NN" + q, b"N" + t, 2, 1)) == "4="
    assert text(oracle.cigar_ops([(0, 0, 3)], b"acn", b"ACR")) == "3="             # case folded, letters outside ACGT are N
    assert oracle.cigar_ops([], q, t) == []
    with pytest.raises(ValueError):
        oracle.cigar_ops([(0, 0, 4), (3, 5, 2)], q, t)


def test_m4_line():
    c = refine.AlignmentCandidate(tAlignedSeq=b"", tAlignedSeqPos=1000, qAlignedSeqPos=10, qPos=0, tPos=0,
                                  blocks=_blocks([(2, 3, 50), (60, 70, 40)]), tStrand=1, score=-420)
    c.pctSimilarity = 87.5
    f = emit.m4_line(c, "m/1/0_100", "chr", q_length=120, t_length=5000, map_qv=254).split()
    assert len(f) == 13
    assert f[:4] == ["m/1/0_100", "chr", "-420", "87.5"]
    assert [int(x) for x in f[4:]] == [0, 12, 110, 120, 1, 1003, 1110, 5000, 254]


def test_m5_line_round_trips_through_the_ingest_hook_and_sam_is_consistent():
    """emit.m5_line is parsed back by tests/ingest_goldens.py (the hook for real `blasr -m 5` output) into the same
    blocks and counts; emit.sam_line agrees with it on lengths, strand and position."""
    import re
    import ingest_goldens as ig
    from blasr_amd import emit, refine
    read = b"TTACGTACGTAAGGCCTTACGATCGATTTGGA"
    window = b"GGACGTACGTAAGCCTTACGAATCGATTTCC"
    blk = np.zeros(3, refine.BLOCK_DT)
    blk[0] = (0, 0, 11); blk[1] = (12, 11, 8); blk[2] = (20, 20, 8)       # one inserted read base, one deleted reference base
    c = refine.AlignmentCandidate(tAlignedSeq=window[2:], qAlignedSeq=read[2:], tAlignedSeqPos=1002, qAlignedSeqPos=2, blocks=blk,
                                  score=-111, nMatch=27, nMismatch=0, nIns=1, nDel=1)
    line = emit.m5_line(c, "read/0", "ref", len(read), 5000, map_qv=30)
    rec = ig.parse_m5(line)
    assert rec is not None and (rec["qStart"], rec["qEnd"], rec["tStart"], rec["tEnd"]) == (2, 30, 1002, 1030)
    q, t, blocks = ig.blocks_from_strings(rec["qAlignedSeq"], rec["tAlignedSeq"])
    assert q == read[2:30] and t == window[2:30] and blocks == [(0, 0, 11), (12, 11, 8), (20, 20, 8)]
    same = sum(read[2 + bq + i] == window[2 + bt + i] for bq, bt, ln in blocks for i in range(ln))
    assert rec["matchPattern"].count("|") == same and rec["matchPattern"].count("*") == 27 - same + 2
    assert (rec["numIns"], rec["numDel"], rec["score"], rec["mapQV"]) == (1, 1, -111, 30)
    sam = emit.sam_line(c, "read/0", "ref", read, 5000, map_qv=30, use_seq_match=True, edit_distance=2).split("\t")
    assert sam[1] == "0" and sam[3] == "1003" and sam[4] == "30" and sam[9] == read.decode()
    ops = [(int(n), o) for n, o in re.findall(r"(\d+)([MIDS=X])", sam[5])]
    assert [o for _, o in ops if o in "SID"] == ["S", "I", "D", "S"] and ops[0] == (2, "S") and ops[-1] == (2, "S")
    assert sum(n for n, o in ops if o == "=") == same and sum(n for n, o in ops if o in "=X") == 27
    assert sum(n for n, o in ops if o in "SI=XM") == len(read)
    assert "AS:i:-111" in sam and "NM:i:2" in sam
    # the same candidate on the reverse strand: flag 16, read reverse-complemented, CIGAR reversed, POS from the far end
    c.tStrand = 1
    rs = emit.sam_line(c, "read/0", "ref", read, 5000, use_seq_match=True).split("\t")
    assert rs[1] == "16" and rs[3] == str(5000 - 1030 + 1)
    assert [(int(n), o) for n, o in re.findall(r"(\d+)([MIDS=X])", rs[5])] == ops[::-1]
    assert rs[9] == read.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1].decode()
