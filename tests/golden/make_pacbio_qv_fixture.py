#!/usr/bin/env python3
"""Generate tests/golden/pacbio_subreads_qv.json: real PacBio subreads with their per-base quality values, taken
from the only read data the reference tree holds with non-trivial quality values,
utils/bam2bax/tests/data/m140905_042212_sidney_c100564852550000001823085912221377_s1_X0.1.subreads.bam (117 unaligned
subreads; tags iq = InsertionQV, dq = DeletionQV, dt = DeletionTag, sq = SubstitutionQV, mq = MergeQV; no
SubstitutionTag).  The 22 subreads of tiny.subreads.sam beside it carry only the constant quality value 0 and the
tag N, so they say nothing about quality-value ranges and are not used.

The BAM is read with gzip + struct (nothing from the file is executed).  There is no reference sequence these reads
belong to and no expected alignment in the reference tree, so every read gets a synthetic target (the read with
seeded random edits) and a guide of exact matches; the expected results are those of the CPU oracles on the costs
refine.ids_profile derives from the real quality values.  Like the other fixtures this pins oracle <-> kernel parity
on realistic cost profiles, not the BLASR binary.

Run in the build container (needs /root/reference); the JSON it writes is committed."""
import difflib
import gzip
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blasr_amd import api, refine, _lib      # noqa: E402  (host-side parameter code only: no device is touched)
from oracle import oracle                    # noqa: E402

BAM = "/root/reference/utils/bam2bax/tests/data/m140905_042212_sidney_c100564852550000001823085912221377_s1_X0.1.subreads.bam"


def read_bam(path):
    d = gzip.open(path, "rb").read()
    assert d[:4] == b"BAM\1"
    p = 8 + struct.unpack_from("<i", d, 4)[0]
    n_ref = struct.unpack_from("<i", d, p)[0]; p += 4
    for _ in range(n_ref):
        p += 4 + struct.unpack_from("<i", d, p)[0] + 4
    out = []
    while p < len(d):
        bs = struct.unpack_from("<i", d, p)[0]; q = p + 4; end = q + bs
        _, _, l_rn, _, _, n_cig, _, l_seq, _, _, _ = struct.unpack_from("<iiBBHHHiiii", d, q); q += 32
        name = d[q:q + l_rn - 1].decode(); q += l_rn + 4 * n_cig
        sb = d[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2 + l_seq
        seq = "".join("=ACMGRSVTWYHKDBN"[(sb[i // 2] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        tags = {}
        while q < end:
            tag, ty = d[q:q + 2].decode(), chr(d[q + 2]); q += 3
            if ty == "Z":
                e = d.index(b"\0", q); tags[tag] = d[q:e].decode(); q = e + 1
            elif ty in "AcC": q += 1
            elif ty in "sS": q += 2
            elif ty in "iIf": q += 4
            elif ty == "B":
                q += 5 + struct.unpack_from("<i", d, q + 1)[0] * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[chr(d[q])]
            else:
                raise SystemExit("unexpected tag type " + ty)
        out.append((name, seq, tags)); p = end
    return out


def synthetic_target(rng, read):
    acgt = "ACGT"
    t = []
    for c in read:
        r = rng.random()
        if r < 0.05:                    # the read has an extra base here
            continue
        if r < 0.08:                    # the read lacks a base here
            t.append(acgt[int(rng.integers(0, 4))])
        t.append(acgt[int(rng.integers(0, 4))] if r > 0.98 else c)
    return "".join(t)


def qv_of(tags, n):
    dec = lambda s: bytes(ord(c) - 33 for c in s)
    return refine.QualityValues(qual=None, insertionQV=dec(tags["iq"]), deletionQV=dec(tags["dq"]), deletionTag=tags["dt"].encode(),
                                substitutionQV=dec(tags["sq"]), substitutionTag=None)


def main():
    recs = [r for r in read_bam(BAM) if 250 <= len(r[1]) <= 1700 and set(r[1]) <= set("ACGT")]
    recs = recs[::4][:16]
    rng = np.random.default_rng(20250218)
    params = refine.MappingParameters(ignoreQualities=False)
    cases = []
    for name, seq, tags in recs:
        tgt = synthetic_target(rng, seq)
        mb = [m for m in difflib.SequenceMatcher(None, seq, tgt, autojunk=False).get_matching_blocks() if m.size >= 11]
        assert mb, name
        q0, t0 = mb[0].a, mb[0].b
        blocks = [[m.a - q0, m.b - t0, m.size] for m in mb]
        ql, tl = blocks[-1][0] + blocks[-1][2], blocks[-1][1] + blocks[-1][2]
        qs, ts = seq[q0:q0 + ql], tgt[t0:t0 + tl]
        prof = refine.ids_profile(seq.encode(), qv_of(tags, len(seq)), params)[q0 * 11:(q0 + ql) * 11]
        exp = {}
        lin = oracle.make_params(ins=params.deletion, dele=params.insertion, indel=params.indel, band=params.guidedAlignBandSize)
        aff = oracle.make_params(ins=params.deletion, dele=params.insertion, indel=params.indel, band=params.bandSize, mode=oracle.MODE_AFFINE,
                                 affine_open=params.affineOpen, affine_extend=params.affineExtend, stats_affine=1)
        for key, P in (("guided_linear", lin), ("guided_affine", aff)):
            res, blk = oracle.align_qv(qs, ts, blocks, P, prof)
            exp[key] = {"result": {a: int(b) for a, b in res.items()}, "blocks": [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in blk]}
        k = refine.compute_drift(np.array([tuple(b) for b in blocks], dtype=_lib.BLOCK_DT)) + 5
        KP = api.make_kband_params(align_type=_lib.KB_GLOBAL, ins=params.insertion, dele=params.deletion)
        res, blk = oracle.kband_align(qs, ts, k, KP, profile=prof)
        exp["kband_global"] = {"k": int(k), "result": {a: int(b) for a, b in res.items()},
                               "blocks": [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in blk]}
        cases.append({"name": name, "read": seq, "iq": tags["iq"], "dq": tags["dq"], "dt": tags["dt"], "sq": tags["sq"], "mq": tags["mq"],
                      "target": tgt, "q0": q0, "t0": t0, "blocks": blocks, "expect": exp})
    with open(os.path.join(ROOT, "tests", "golden", "pacbio_subreads_qv.json"), "w") as f:
        json.dump({"made_by": "tests/golden/make_pacbio_qv_fixture.py: reads and quality values from the reference's "
                              "utils/bam2bax/tests/data/*.subreads.bam; targets synthetic; expected results from oracle/ "
                              "(parity with BLASR unpinned)", "cases": cases}, f)
    print(len(cases), "cases,", sum(len(c["read"]) for c in cases), "bases")


if __name__ == "__main__":
    main()
