"""Quality-value costs inside the guided DP (gba_batch.q_profile: GuidedAlign with the IDS scorer,
/root/reference/iblasr/BlasrUtilsImpl.hpp:705-717), through the C-ABI against oracle/gba_oracle.c's
gbo_align_qv, bit for bit: host and device batches, narrow and wide bands, the k-band geometry, packed reads."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _profile(rng, batch):
    """Random costs for every byte of the query arena, a few of them outside the +-127 the kernels take."""
    n = len(batch["q_arena"])
    p = np.zeros((n, 11), np.int16)
    p[:, :5] = rng.integers(1, 40, (n, 5))
    code = np.full(256, 4, np.int64)
    for i, c in enumerate(b"ACGT"):
        code[c] = i; code[c + 32] = i
    p[np.arange(n), code[batch["q_arena"]]] = rng.integers(-9, 1, n)
    p[:, 5] = rng.integers(0, 25, n)
    p[:, 6:] = rng.integers(0, 25, (n, 5))
    wild = rng.random(n) < 0.01
    p[wild, 5] = 300
    p[wild, 7] = -200
    return p.reshape(-1)


def _check(oracle, batch, prof, OP, pairs, blocks):
    n = int(batch["n_pairs"])
    from blasr_amd import synth
    for p in range(n):
        q, t, blk = synth.pair(batch, p)
        o = int(batch["q_off"][p])
        want, wb = oracle.align_qv(q, t, blk, OP, prof[o * 11:(o + len(q)) * 11])
        g = pairs[p]
        assert int(g["status"]) == want["status"], (p, int(g["status"]), want["status"])
        if want["status"] != 0:
            continue
        for f in util.FIELDS:
            assert int(g[f]) == int(want[f]), (f, p, int(g[f]), int(want[f]))
        kb = blocks[int(g["blk_off"]):int(g["blk_off"]) + int(g["n_blk"])]
        assert (kb == wb).all(), p


@pytest.mark.parametrize("family", ["lane-per-pair", "wave-per-pair"])
@pytest.mark.parametrize("band", [10, 120])
@pytest.mark.parametrize("tie", [0, 1])
def test_profile_costs_in_the_guided_dp(oracle, synth, monkeypatch, band, tie, family):
    """Both kernel families: one lane per pair (k_fill<false, true>; with band 120 only the pairs narrow enough for a
    lane class) and one wave per pair (k_fill_wide<.., true>)."""
    import torch
    from blasr_amd import api, _lib
    rng = np.random.default_rng(5 + band + tie)
    batch = synth.compact_blocks(synth.generate(160, read_len=900, len_jitter=500, seed=31 + band))
    prof = _profile(rng, batch)
    kw = dict(band=band, tie_break=tie, ins=5, dele=4)
    monkeypatch.setenv("GBA_WAVE_PAIRS", "1" if family == "lane-per-pair" else "4000000000")
    ctx = api.Context(0, timing=True)
    ctx.set_scoring(api.make_scoring(**kw))
    OP = oracle.make_params(**kw)
    with_prof = dict(batch, q_profile=prof)
    pairs, blocks, rc = ctx.refine_host(with_prof)
    assert rc == 0 and int((pairs["status"] == 0).sum()) == 160
    wide = int(ctx.timing()["wide_pairs"])
    assert wide == 160 if family == "wave-per-pair" or band == 120 else wide == 0, wide
    _check(oracle, batch, prof, OP, pairs, blocks)
    # the costs matter: the same batch without them scores differently
    plain, _, _ = ctx.refine_host(batch)
    assert (plain["dp_score"] != pairs["dp_score"]).any()
    # device-resident batch, and the reads 2 bits per base
    d = api.to_device(with_prof)
    out_pairs = torch.zeros(160 * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
    out_blocks = torch.zeros((len(blocks) + 8, 3), dtype=torch.int32, device="cuda:0")
    rc, needed = ctx.refine_device(d, out_pairs, out_blocks)
    torch.cuda.synchronize()
    assert rc == 0 and out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT).tobytes() == pairs.tobytes()
    pk = api.pack_reads_2bit(batch)
    pk["q_profile"] = prof
    p2, b2, rc = ctx.refine_host(pk)
    assert rc == 0 and p2.tobytes() == pairs.tobytes() and b2.tobytes() == blocks.tobytes()
    ctx.close()


@pytest.mark.parametrize("k,family", [(150, "wave-per-pair"), (40, "lane-per-pair")])
def test_profile_with_the_kband_geometry(oracle, synth, monkeypatch, k, family):
    from blasr_amd import api, _lib
    rng = np.random.default_rng(2)
    batch = synth.compact_blocks(synth.generate(40, read_len=500, len_jitter=100 if k == 150 else 30, seed=9))
    prof = _profile(rng, batch)
    kw = dict(mode=_lib.MODE_KBAND, kband_k=k, ins=6, dele=6)
    monkeypatch.setenv("GBA_WAVE_PAIRS", "1" if family == "lane-per-pair" else "4000000000")
    ctx = api.Context(0)
    ctx.set_scoring(api.make_scoring(**kw))
    pairs, blocks, rc = ctx.refine_host(dict(batch, q_profile=prof))
    assert rc == 0
    _check(oracle, batch, prof, oracle.make_params(**kw), pairs, blocks)
    ctx.close()


@pytest.mark.parametrize("family", ["lane-per-pair", "wave-per-pair"])
@pytest.mark.parametrize("band,oe", [(8, (4, 1)), (16, (10, 0)), (150, (3, 2)), (40, (0, 7))])
def test_profile_costs_with_affine_gaps(oracle, synth, monkeypatch, band, oe, family):
    """AffineGuidedAlign with the IDS scorer (/root/reference/iblasr/BlasrUtilsImpl.hpp:707-710): per-base costs for
    substitution, insertion and the one-base deletion (taken within +-63), constant gap open / extend; narrow and wide
    rows, both tie-breaks, an extend cost above and below the per-base deletion costs, host and device batches; one
    lane per pair (k_fill<true, true>, the pairs narrow enough for a lane class) and one wave per pair."""
    import torch
    from blasr_amd import api, _lib
    rng = np.random.default_rng(17 + band)
    batch = synth.compact_blocks(synth.generate(96, read_len=800, len_jitter=500, seed=3 + band))
    prof = _profile(rng, batch)
    monkeypatch.setenv("GBA_WAVE_PAIRS_AFFINE", "1" if family == "lane-per-pair" else "4000000000")
    ctx = api.Context(0, timing=True)
    for tie in (0, 1):
        kw = dict(mode=_lib.MODE_AFFINE, band=band, tie_break=tie, ins=5, dele=4, affine_open=oe[0], affine_extend=oe[1],
                  stats_affine=1, stats_affine_open=2, stats_affine_extend=1)
        ctx.set_scoring(api.make_scoring(**kw))
        with_prof = dict(batch, q_profile=prof)
        pairs, blocks, rc = ctx.refine_host(with_prof)
        assert rc == 0 and int((pairs["status"] == 0).sum()) == 96
        wide = int(ctx.timing()["wide_pairs"])
        assert wide == 96 if family == "wave-per-pair" or band > 40 else wide < 48, wide        # with a profile affine batches have a 128-slot lane class
        _check(oracle, batch, prof, oracle.make_params(**kw), pairs, blocks)
    plain, _, _ = ctx.refine_host(batch)
    assert (plain["dp_score"] != pairs["dp_score"]).any()
    d = api.to_device(with_prof)
    out_pairs = torch.zeros(96 * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
    out_blocks = torch.zeros((len(blocks) + 8, 3), dtype=torch.int32, device="cuda:0")
    rc, needed = ctx.refine_device(d, out_pairs, out_blocks)
    torch.cuda.synchronize()
    assert rc == 0 and out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT).tobytes() == pairs.tobytes()
    ctx.close()


def test_affine_profile_that_spells_out_the_matrix_gives_the_plain_affine_results(synth, monkeypatch):
    """A profile whose costs are the matrix row of each base, ins and del: the composed-map row scan of the
    quality-value kernel reproduces the two-slope scan of the plain affine kernel (no oracle involved), reads of 6 kb."""
    from blasr_amd import api, _lib
    n = 256
    host = synth.compact_blocks(synth.generate(n, read_len=6000, len_jitter=1000, seed=77))
    kw = dict(mode=_lib.MODE_AFFINE, match=-5, mismatch=6, ins=5, dele=4, band=16, affine_open=6, affine_extend=2, stats_affine=1)
    sc = api.make_scoring(**kw)
    code = np.full(256, 4, np.int64)
    for i, c in enumerate(b"ACGT"):
        code[c] = i; code[c + 32] = i
    sub = np.array([[sc.sub[a][b] for b in range(5)] for a in range(5)], np.int16)
    qc = code[host["q_arena"]]
    prof = np.zeros((len(qc), 11), np.int16)
    prof[:, :5] = sub[qc]
    prof[:, 5] = kw["ins"]
    prof[:, 6:] = kw["dele"]
    for wave_pairs in ("4000000000", "1"):          # one wave per pair, then the lane classes
        monkeypatch.setenv("GBA_WAVE_PAIRS_AFFINE", wave_pairs)
        ctx = api.Context(0)
        ctx.set_scoring(sc)
        a = ctx.refine_host(host)
        b = ctx.refine_host(dict(host, q_profile=prof.reshape(-1)))
        assert a[2] == 0 and b[2] == 0 and int((a[0]["status"] == 0).sum()) == n
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
        ctx.close()


@pytest.mark.parametrize("affine", [False, True])
def test_refine_alignments_with_quality_values(oracle, synth, affine):
    """refine.RefineAlignments(..., qv=...) with -useQuality: the IDS scorer's costs (refine.ids_profile) drive the
    guided DP (with -affineAlign: AffineGuidedAlign, gap costs params.affineOpen / affineExtend); against the oracle on
    the same slices and costs."""
    from blasr_amd import api, refine
    rng = np.random.default_rng(11)
    batch = synth.generate(6, read_len=700, len_jitter=200, seed=123)
    ctx = api.Context(0)
    params = refine.MappingParameters(ignoreQualities=False, affineAlign=affine)
    ACGT = np.frombuffer(b"ACGT", np.uint8)
    for i in range(6):
        q, t, blk = synth.pair(batch, i)
        read, n = bytes(q), len(q)
        qv = refine.QualityValues(qual=bytes(rng.integers(1, 40, n, dtype=np.uint8)), insertionQV=bytes(rng.integers(1, 20, n, dtype=np.uint8)),
                                  deletionQV=bytes(rng.integers(1, 20, n, dtype=np.uint8)), deletionTag=rng.choice(np.frombuffer(b"ACGTN", np.uint8), n).tobytes(),
                                  substitutionQV=bytes(rng.integers(1, 30, n, dtype=np.uint8)), substitutionTag=rng.choice(ACGT, n).tobytes())
        q0, t0 = int(blk[0]["q_pos"]), int(blk[0]["t_pos"])
        rel = blk.copy(); rel["q_pos"] -= np.uint32(q0); rel["t_pos"] -= np.uint32(t0)
        c = refine.AlignmentCandidate(tAlignedSeq=bytes(t), qPos=q0, tPos=t0, blocks=rel)
        refine.RefineAlignments(read, [c], params, ctx, qv=qv)
        ql = int(rel[-1]["q_pos"]) + int(rel[-1]["length"]); tl = int(rel[-1]["t_pos"]) + int(rel[-1]["length"])
        prof = refine.ids_profile(read, qv, params)
        OP = oracle.make_params(ins=params.deletion, dele=params.insertion, indel=params.indel, band=params.guidedAlignBandSize)
        if affine:
            OP = oracle.make_params(ins=params.deletion, dele=params.insertion, indel=params.indel, band=params.bandSize,
                                    mode=oracle.MODE_AFFINE, affine_open=params.affineOpen, affine_extend=params.affineExtend,
                                    stats_affine=1, stats_affine_open=refine.STATS_SCORER_AFFINE_OPEN,
                                    stats_affine_extend=refine.STATS_SCORER_AFFINE_EXTEND)
        want, wb = oracle.align_qv(read[q0:q0 + ql], bytes(t)[t0:t0 + tl], rel, OP, prof[q0 * 11:(q0 + ql) * 11])
        assert c.status == 0 and want["status"] == 0
        assert (c.score, c.nCells, c.nMatch, len(c.blocks)) == (want["score"], want["ncells"], want["n_match"], want["n_blk"])
        assert (c.blocks == wb).all()
    ctx.close()


def test_cpp_shim_with_quality_values(oracle, synth):
    """refine_demo qv: gba::IdsProfile + RefineBatch::Add(read, cand, &profile) give what the Python mirror gives."""
    import os
    import subprocess
    from blasr_amd import api, build, refine
    exe = build.HOST_DEMO
    assert os.path.exists(exe), "run python -m blasr_amd.build"
    rng = np.random.default_rng(21)
    batch = synth.generate(5, read_len=600, len_jitter=200, seed=55)
    ctx = api.Context(0)
    params = refine.MappingParameters(ignoreQualities=False, sortRefinedAlignments=False)
    ACGT = np.frombuffer(b"ACGT", np.uint8)
    lines, want = ["5"], []
    enc = lambda b: bytes(x + 33 for x in b).decode()
    for i in range(5):
        q, t, blk = synth.pair(batch, i)
        read, n = bytes(q), len(q)
        q0, t0 = int(blk[0]["q_pos"]), int(blk[0]["t_pos"])
        rd, tg = read[q0:], bytes(t)[t0:]            # the demo's candidates start at (0, 0)
        n = len(rd)
        qv = refine.QualityValues(qual=bytes(rng.integers(1, 40, n, dtype=np.uint8)), insertionQV=bytes(rng.integers(1, 20, n, dtype=np.uint8)),
                                  deletionQV=bytes(rng.integers(1, 20, n, dtype=np.uint8)), deletionTag=rng.choice(np.frombuffer(b"ACGTN", np.uint8), n).tobytes(),
                                  substitutionQV=bytes(rng.integers(1, 30, n, dtype=np.uint8)), substitutionTag=rng.choice(ACGT, n).tobytes())
        rel = blk.copy(); rel["q_pos"] -= np.uint32(q0); rel["t_pos"] -= np.uint32(t0)
        c = refine.AlignmentCandidate(tAlignedSeq=tg, blocks=rel.copy())
        refine.RefineAlignments(rd, [c], params, ctx, qv=qv)
        want.append(c)
        lines.append("%s %s %d %s %s %s %s %s %s" % (rd.decode(), tg.decode(), len(rel), " ".join("%d %d %d" % tuple(int(v) for v in b) for b in rel),
                                                      enc(qv.insertionQV), enc(qv.deletionQV), qv.deletionTag.decode(), enc(qv.substitutionQV), qv.substitutionTag.decode()))
    out = subprocess.run([exe, "qv"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    for row, c in zip(out.stdout.strip().split("\n"), want):
        v = [int(x) for x in row.split()]
        assert v[:6] == [c.status, c.score, c.nCells, c.qPos, c.tPos, len(c.blocks)]
        assert v[6:] == [int(x) for b in c.blocks for x in b]
    ctx.close()


def test_a_profile_that_spells_out_the_matrix_gives_the_plain_results(synth, monkeypatch):
    """Size-independent property at the benchmark's read length: a profile whose costs are the matrix row of each
    base, ins and del reproduces, on the lane-per-pair kernel with per-row cost tables and on the wave-per-pair kernels
    with the prefix-sum deletion chain, exactly what the lane-per-pair kernels give for the matrix scorer (8 192 pairs
    of 10 kb, no oracle involved)."""
    import torch
    from blasr_amd import api, _lib
    n = 8192
    host = synth.compact_blocks(synth.preset("c2_10kb_linear", n)[0])
    kw = dict(match=-5, mismatch=6, ins=5, dele=4, band=10)
    sc = api.make_scoring(**kw)
    code = np.full(256, 4, np.int64)
    for i, c in enumerate(b"ACGT"):
        code[c] = i; code[c + 32] = i
    sub = np.array([[sc.sub[a][b] for b in range(5)] for a in range(5)], np.int16)
    qc = code[host["q_arena"]]
    prof = np.zeros((len(qc), 11), np.int16)
    prof[:, :5] = sub[qc]
    prof[:, 5] = kw["ins"]
    prof[:, 6:] = kw["dele"]
    d = api.to_device(host)
    dp = dict(d, q_profile=torch.from_numpy(prof.reshape(-1)).to("cuda:0"))
    cap = int(host["q_len"].astype(np.int64).sum() // 4) + n
    outs = []
    # lane classes even at this batch size: the plain batch, the batch with the profile; then the latter one wave per pair
    for wave_pairs, b in (("1000", d), ("1000", dp), ("4000000000", dp)):
        monkeypatch.setenv("GBA_WAVE_PAIRS", wave_pairs)
        ctx = api.Context(0, timing=True)
        ctx.set_scoring(sc)
        op = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
        ob = torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0")
        rc, needed = ctx.refine_device(b, op, ob)
        torch.cuda.synchronize()
        assert rc == 0
        outs.append((op.cpu().numpy().tobytes(), ob[:needed].cpu().numpy().tobytes(), int(ctx.timing()["wide_pairs"])))
        ctx.close()
    assert outs[0][:2] == outs[1][:2] == outs[2][:2]
    assert outs[0][2] < n // 2 and outs[1][2] == outs[0][2] and outs[2][2] == n


@pytest.mark.parametrize("family", ["lane-per-pair", "wave-per-pair"])
def test_score_range_guard_counts_the_profile_costs(oracle, synth, monkeypatch, family):
    """A pair near the row limit whose profile holds the largest costs the kernels take: its scores would leave the
    cell word (rows x (128 + 127) + columns x 127, or rows x 127 + columns x (128 + 127), reaches 2^24), so it comes back
    TOO_LONG instead of a wrapped score;
    a shorter one with the same costs fits and agrees with the oracle."""
    from blasr_amd import api, _lib
    monkeypatch.setenv("GBA_WAVE_PAIRS", "1" if family == "lane-per-pair" else "4000000000")
    ctx = api.Context(0)
    kw = dict(band=10, ins=5, dele=4)
    ctx.set_scoring(api.make_scoring(**kw))
    for length, want_status in ((47000, _lib.PAIR_TOO_LONG), (30000, _lib.PAIR_OK)):
        batch = synth.compact_blocks(synth.generate(1, read_len=length, len_jitter=0, seed=length))
        n = len(batch["q_arena"])
        prof = np.full((n, 11), 127, np.int16).reshape(-1)
        plain, _, rc = ctx.refine_host(batch)
        assert rc == 0 and int(plain[0]["status"]) == _lib.PAIR_OK          # the matrix scorer's costs fit either way
        pairs, blocks, rc = ctx.refine_host(dict(batch, q_profile=prof))
        assert rc == 0 and int(pairs[0]["status"]) == want_status, (length, int(pairs[0]["status"]))
        if want_status == _lib.PAIR_OK:
            _check(oracle, batch, prof, oracle.make_params(**kw), pairs, blocks)
    ctx.close()
