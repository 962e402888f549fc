"""The read-level harness (blasr_amd/host/read_harness.cpp: candidates of many reads through ONE from-sequences submit, then
per read the reference's post-refine steps and -m 4 lines, /root/reference/iblasr/BlasrAlignImpl.hpp:358-383,
/root/reference/Blasr.cpp:419-430) against the same pipeline built from the Python mirrors, line for line."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_harness_lines_equal_the_python_pipeline(synth, tmp_path):
    from blasr_amd import api, build, emit, mapqv, refine, _lib
    exe = build.READ_HARNESS
    assert os.path.exists(exe), "run python -m blasr_amd.build"
    reads, K, read_len, jitter, ref_len, seed = 48, 4, 1500, 600, 400_000, 77
    dump = str(tmp_path / "m4.txt")
    out = subprocess.run([exe, "--reads", str(reads), "--cands", str(K), "--read-len", str(read_len), "--jitter", str(jitter), "--ref-len", str(ref_len),
                          "--n-frac", "0.01", "--batches", "2", "--seed", str(seed), "--threads", "3", "--dump", dump],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    assert stats["reads_per_sec"] > 0 and stats["reads_per_batch"] == reads and stats["alignments_printed"] >= 2 * reads * 0.9
    got = open(dump).read().splitlines()

    # the same batch (the harness's second buffer: seed + 1, reads numbered from `reads` on) through the Python mirrors
    params = refine.MappingParameters()
    ref = synth.reference(ref_len, synth.REF_SEED, 0.01)
    batch = synth.generate(reads * K, read_len=read_len, len_jitter=jitter, seed=seed + 1, decoy_every=K, share_reads=True, ref=ref)
    w = synth.as_windows(batch)
    seqs = {k: v for k, v in w.items() if k not in ("blocks", "blk_off", "n_blk")}
    sc = refine.scoring_from_params(params)
    sc.n_extra = 1
    for a in range(5):
        for b in range(5):
            sc.extra[0].sub[a][b] = 0 if (a == 4 or b == 4 or a == b) else 15
    sc.extra[0].ins, sc.extra[0].dele = params.insertion, params.deletion
    sc.extra[0].affine, sc.extra[0].affine_open, sc.extra[0].affine_extend = 0, params.affineOpen, params.affineExtend
    ctx = api.Context(0)
    try:
        ctx.set_scoring(sc)
        ctx.set_reference(ref)
        ctx.set_guide_producer(api.make_sdp_params(tuple_size=params.sdpTupleSize, ins=params.sdpIns, dele=params.sdpDel, match=-params.match,
                                                   max_fragments=params.sdpMaxFragments))
        pairs, rc = ctx.refine_host_intervals(seqs)
        assert rc == 0
    finally:
        ctx.close()
    want = []
    for r in range(reads):
        cands, raws = [], {}
        for k in range(K):
            p = r * K + k
            R = pairs[p]
            rcs = int(w["t_rc"][p])
            tpos = ref_len - int(w["t_off"][p]) - int(w["t_len"][p]) if rcs else int(w["t_off"][p])
            c = refine.AlignmentCandidate(tAlignedSeq=b"", tAlignedSeqPos=tpos, tStrand=rcs, status=int(R["status"]))
            c.pair = p
            if c.status == 0:
                c.score, c.nBlocks = int(R["score"]), int(R["n_blk"])
                cols = int(R["n_match"]) + int(R["n_mismatch"]) + int(R["n_ins"]) + int(R["n_del"])
                c.pctSimilarity = float(np.float32(int(R["n_match"]) * 2.0 / (2.0 * cols) * 100.0)) if cols else 0.0
                c.qPos, c.tPos = int(R["q_pos"]), int(R["t_pos"])
                c.qInterval = (int(R["q_start"]), int(R["q_end"]))
                c.tInterval = (tpos + int(R["t_start"]), tpos + int(R["t_end"]))
            raws[id(c)] = int(R["extra_score"][0])
            cands.append(c)
        cands.sort(key=lambda c: c.score)
        refine.RemoveLowQualityAlignments(cands, params)
        refine.RemoveOverlappingAlignments(cands, params)
        mapqv.store_map_qvs(cands, [raws[id(c)] for c in cands], params.minFractionToBeConsideredOverlapping, params.nCandidates)
        for c in cands[:params.nBest]:
            want.append(emit.m4_line(c, "read%d" % (reads + r), "ref", int(w["q_len"][c.pair]), ref_len, map_qv=c.mapQV))
    assert len(got) == len(want) == stats["alignments_printed"] // 2
    assert got == want
    # every read keeps its true candidate (the first of its K); the decoys are filtered out or rank behind it
    first = [l for l in got if l.split()[0] == "read%d" % reads]
    assert len(first) >= 1 and int(first[0].split()[2]) < params.maxScore
