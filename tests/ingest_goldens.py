#!/usr/bin/env python3
"""Check the oracle against alignments printed by a real BLASR run (`blasr ... -m 5`).

No BLASR output exists in this repository: the reference's alignment library and all of its test
data are absent, so the oracle's tie-breaks and band rules are pinned only by the reference's
call-site contract.  This hook is how that changes the day somebody has a blasr binary: run it on
any reads / reference with `-m 5`, hand the output to this script, and every record is re-aligned
by the oracle with the record's own gap-free blocks as the guide (the same thing RefineAlignment
does with the SDP blocks, /root/reference/iblasr/BlasrUtilsImpl.hpp:676-772) and compared with
what BLASR printed: the path (blocks), nMatch / nMismatch / nIns / nDel and the score.

The `-m 5` record layout (CompareSequencesParsable, /root/reference/iblasr/RegisterBlasrOptions.h:276)
is taken from BLASR's manual, space separated:
  qName qLength qStart qEnd qStrand tName tLength tStart tEnd tStrand score numMatch numMismatch
  numIns numDel mapQV qAlignedSeq matchPattern tAlignedSeq
with '-' for gap columns in the two aligned strings.

A path BLASR found inside the band around the SDP guide is also inside the band around itself, so
the oracle's DP score on the re-alignment can only be equal or better; `summary["dp_worse"]` counts
violations of that, `summary["path_equal"]` the records whose path is reproduced exactly.

usage: python tests/ingest_goldens.py alignments.m5 [--affine] [--band N] [--tie-break 0|1] ...
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

M5_FIELDS = ("qName", "qLength", "qStart", "qEnd", "qStrand", "tName", "tLength", "tStart", "tEnd", "tStrand",
             "score", "numMatch", "numMismatch", "numIns", "numDel", "mapQV", "qAlignedSeq", "matchPattern",
             "tAlignedSeq")
_INT = {"qLength", "qStart", "qEnd", "tLength", "tStart", "tEnd", "score", "numMatch", "numMismatch", "numIns",
        "numDel", "mapQV"}


def parse_m5(line):
    """One `-m 5` line -> dict, or None for blank / header lines."""
    parts = line.split()
    if len(parts) != len(M5_FIELDS) or parts[0] == "qName":
        return None
    rec = dict(zip(M5_FIELDS, parts))
    for k in _INT:
        rec[k] = int(rec[k])
    if not (len(rec["qAlignedSeq"]) == len(rec["matchPattern"]) == len(rec["tAlignedSeq"])):
        raise ValueError("aligned strings of %s differ in length" % rec["qName"])
    return rec


def blocks_from_strings(q_aln, t_aln):
    """Gapped strings -> (q bytes, t bytes, [(q_pos, t_pos, length)...]): maximal gap-free runs."""
    q, t, blocks = bytearray(), bytearray(), []
    run = 0
    for a, b in zip(q_aln, t_aln):
        ga, gb = a == "-", b == "-"
        if ga and gb:
            continue
        if not ga and not gb:
            if run == 0:
                blocks.append([len(q), len(t), 0])
            run += 1
            blocks[-1][2] = run
        else:
            run = 0
        if not ga:
            q.append(ord(a))
        if not gb:
            t.append(ord(b))
    return bytes(q), bytes(t), [tuple(b) for b in blocks]


def strings_from_blocks(q, t, blocks):
    """Inverse of blocks_from_strings for a global path: query-only stretches before target-only ones."""
    qs, ts, qp, tp = [], [], 0, 0
    q, t = q.decode("latin1"), t.decode("latin1")
    for bq, bt, ln in list(blocks) + [(len(q), len(t), 0)]:
        qs.append(q[qp:bq] + "-" * (bt - tp)); ts.append("-" * (bq - qp) + t[tp:bt])
        qs.append(q[bq:bq + ln]); ts.append(t[bt:bt + ln])
        qp, tp = bq + ln, bt + ln
    return "".join(qs), "".join(ts)


def format_m5(name, q, t, blocks, score, counts, t_name="ref", strand="+"):
    """A `-m 5` line for a global alignment of q against t along `blocks` (used by the round-trip test)."""
    qa, ta = strings_from_blocks(q, t, blocks)
    pat = "".join("|" if a == b and a != "-" else "*" for a, b in zip(qa.upper(), ta.upper()))
    n_match, n_mismatch, n_ins, n_del = counts
    return " ".join(str(x) for x in (name, len(q), 0, len(q), "+", t_name, len(t), 0, len(t), strand, score,
                                     n_match, n_mismatch, n_ins, n_del, 254, qa, pat, ta))


def check_records(records, oracle, **scoring):
    """Re-align every record with the oracle, guide = the record's own blocks.  Returns (summary, rows)."""
    import util
    P = oracle.make_params(**scoring)
    trip = [blocks_from_strings(r["qAlignedSeq"], r["tAlignedSeq"]) for r in records]
    keep = [i for i, (q, t, b) in enumerate(trip) if b]
    summary = {"records": len(records), "aligned": 0, "path_equal": 0, "counts_equal": 0, "score_equal": 0,
               "dp_worse": 0, "skipped": len(records) - len(keep)}
    rows = []
    if not keep:
        return summary, rows
    # the guide convention puts the first block at target 0 and the query at the guide's extent
    # (/root/reference/iblasr/BlasrUtilsImpl.hpp:693-703), so trim leading target-only columns
    pairs = []
    for i in keep:
        q, t, b = trip[i]
        t0 = b[0][1]
        pairs.append((q, t[t0:], [(bq, bt - t0, ln) for bq, bt, ln in b]))
    batch = util.make_batch(pairs)
    res, blks = oracle.align_batch(batch, P)
    for k, i in enumerate(keep):
        r, o, ob = records[i], res[k], blks[k]
        if int(o["status"]) != 0:
            summary["skipped"] += 1
            continue
        summary["aligned"] += 1
        want = pairs[k][2]
        got = [(int(b["q_pos"]) + int(o["q_pos"]), int(b["t_pos"]) + int(o["t_pos"]), int(b["length"])) for b in ob]
        path_eq = got == want
        cnt_eq = (int(o["n_match"]), int(o["n_mismatch"]), int(o["n_ins"]), int(o["n_del"])) == \
                 (r["numMatch"], r["numMismatch"], r["numIns"], r["numDel"])
        score_eq = int(o["score"]) == r["score"]
        # DP score of BLASR's own path under the alignment scorer, from its columns
        own = _path_dp_score(pairs[k][0], pairs[k][1], want, P, oracle)
        worse = own is not None and int(o["dp_score"]) > own
        summary["path_equal"] += path_eq; summary["counts_equal"] += cnt_eq
        summary["score_equal"] += score_eq; summary["dp_worse"] += worse
        rows.append({"qName": r["qName"], "path_equal": path_eq, "counts_equal": cnt_eq, "score_equal": score_eq,
                     "oracle_score": int(o["score"]), "m5_score": r["score"], "oracle_dp": int(o["dp_score"]),
                     "m5_path_dp": own})
    return summary, rows


def _path_dp_score(q, t, blocks, P, oracle):
    """Linear-gap DP cost of walking `blocks` globally (None for affine scoring)."""
    if P.mode == oracle.MODE_AFFINE:
        return None
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
    s, qp, tp = 0, 0, 0
    for bq, bt, ln in list(blocks) + [(len(q), len(t), 0)]:
        s += (bq - qp) * P.ins + (bt - tp) * P.dele
        for k in range(ln):
            s += P.sub[code.get(q[bq + k], 4)][code.get(t[bt + k], 4)]
        qp, tp = bq + ln, bt + ln
    return s


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("m5")
    ap.add_argument("--affine", action="store_true")
    ap.add_argument("--band", type=int, default=None)
    ap.add_argument("--tie-break", type=int, default=None)
    ap.add_argument("--match", type=int, default=None)
    ap.add_argument("--mismatch", type=int, default=None)
    ap.add_argument("--insertion", type=int, default=None, help="blasr -insertion (default 4)")
    ap.add_argument("--deletion", type=int, default=None, help="blasr -deletion (default 5)")
    ap.add_argument("--show", type=int, default=10, help="print this many differing records")
    a = ap.parse_args()
    from oracle import oracle
    kw = {}
    if a.affine:
        kw.update(mode=1, band=16, stats_affine=1)
    for name, val in (("band", a.band), ("tie_break", a.tie_break), ("match", a.match), ("mismatch", a.mismatch)):
        if val is not None:
            kw[name] = val
    # the distance scorer is built as (matrix, params.deletion, params.insertion) and read as
    # ctor(ins, del): /root/reference/iblasr/BlasrUtilsImpl.hpp:630-631
    if a.deletion is not None:
        kw["ins"] = a.deletion
    if a.insertion is not None:
        kw["dele"] = a.insertion
    with open(a.m5) as f:
        records = [r for r in (parse_m5(line) for line in f) if r]
    summary, rows = check_records(records, oracle, **kw)
    print(summary)
    shown = 0
    for r in rows:
        if not (r["path_equal"] and r["counts_equal"] and r["score_equal"]) and shown < a.show:
            print(r); shown += 1
    sys.exit(0 if summary["aligned"] and summary["path_equal"] == summary["aligned"] and not summary["dp_worse"] else 1)


if __name__ == "__main__":
    main()
