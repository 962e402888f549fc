"""GPU parity tests: everything goes through the C-ABI (libgba.so) and is compared bit for bit
with the CPU oracle.  Parity with the BLASR binary itself is unpinned (see oracle/gba_oracle.c)."""
import json
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def api():
    from blasr_amd import api as a
    return a


@pytest.fixture(scope="module", params=["lane-per-pair", "wave-per-pair"])
def ctx(api, request):
    """Every test that takes this context runs twice: with batches of any size on the lane-per-pair
    kernels (rows too wide for them excepted), and with every batch on the wave-per-pair kernels
    (the library itself sends batches too small to fill the GPU there; the narrow affine wave class four pairs per wave,
    which the library does for large batches only: the one-pair-per-wave instance is what tests/test_gpu_qv_guided.py, the
    randomised driver and every other test file's small affine batches run)."""
    old = {k: os.environ.get(k) for k in ("GBA_WAVE_PAIRS", "GBA_WAVE_PAIRS_AFFINE", "GBA_QUAD_AFFINE_MIN", "GBA_TB_WAVE_MAX", "GBA_TB_FEW_MAX")}
    v = "0" if request.param == "lane-per-pair" else "4000000000"
    os.environ["GBA_WAVE_PAIRS"] = os.environ["GBA_WAVE_PAIRS_AFFINE"] = v
    os.environ["GBA_QUAD_AFFINE_MIN"] = "0"
    # the narrow wave class walked back by the lane-per-pair traceback, as in large batches; small batches on a default
    # context (the other test files, the randomised driver) are walked back one wave per pair
    os.environ["GBA_TB_WAVE_MAX"] = "0"
    # affine pairs walked back with the record window of full batches (small batches on a default context take the large
    # one: test_affine_lane_traceback_windows_agree, the other test files, the randomised driver)
    os.environ["GBA_TB_FEW_MAX"] = "0"
    c = api.Context(0, timing=True)
    for k, o in old.items():
        if o is None:
            del os.environ[k]
        else:
            os.environ[k] = o
    yield c
    c.close()


def _run(ctx, api, oracle, batch, expect_status=None, **kw):
    ctx.set_scoring(api.make_scoring(**kw))
    pairs, blocks, rc = ctx.refine_host(batch)
    assert rc == 0
    util.compare_batch(oracle, batch, oracle.make_params(**kw), pairs, blocks, expect_status)
    return pairs, blocks


@pytest.mark.parametrize("fixture", ["micro.json", "synthetic_small.json"])
def test_golden_fixtures(ctx, api, oracle, fixture):
    cases = json.load(open(os.path.join(GOLD, fixture)))["cases"]
    by_params = {}
    for c in cases:
        by_params.setdefault(json.dumps(c["params"], sort_keys=True), []).append(c)
    for pj, cs in by_params.items():
        kw = json.loads(pj)
        batch = util.make_batch([(c["q"], c["t"], c["blocks"]) for c in cs])
        ctx.set_scoring(api.make_scoring(**kw))
        pairs, blocks, rc = ctx.refine_host(batch)
        assert rc == 0
        for c, k in zip(cs, pairs):
            e = c["expect"]["result"]
            assert int(k["status"]) == e["status"], c["name"]
            if e["status"] != 0:
                continue
            for f in util.FIELDS:
                assert int(k[f]) == e[f], (c["name"], f)
            kb = blocks[int(k["blk_off"]):int(k["blk_off"]) + int(k["n_blk"])]
            assert [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in kb] == c["expect"]["blocks"], c["name"]


def test_c1_plumbing(ctx, api, oracle, synth):
    batch, cfg = synth.preset("c1_ecoli_plumbing", 300)
    _run(ctx, api, oracle, batch)


def test_c2_10kb(ctx, api, oracle, synth):
    batch, cfg = synth.preset("c2_10kb_linear", 192)
    pairs, _ = _run(ctx, api, oracle, batch)
    assert int((pairs["status"] == 0).sum()) == 192


def test_c4_decoys_and_n_runs(ctx, api, oracle, synth):
    ref = synth.reference(3_000_000, seed=9, n_frac=0.02, n_run=1000)
    batch = synth.generate(200, read_len=4000, seed=4000, decoy_every=10, ref=ref)
    _run(ctx, api, oracle, batch)


def test_wide_band(ctx, api, oracle, synth):
    batch = synth.generate(24, read_len=3000, p_ins=0.11, p_del=0.06, p_sub=0.03, seed=5000)
    _run(ctx, api, oracle, batch, band=40)
    _run(ctx, api, oracle, batch, band=100)
    _run(ctx, api, oracle, batch, band=200)          # rows of ~400 cells: the wave-per-pair kernel


def test_c5_band250(ctx, api, oracle, synth):
    # rows of a band-250 guide have up to 501 cells plus the shift: all of them go to the
    # wave-per-pair kernel, none is refused
    batch = synth.generate(8, read_len=6000, p_ins=0.11, p_del=0.06, p_sub=0.03, seed=5001)
    pairs, _ = _run(ctx, api, oracle, batch, band=250)
    assert int((pairs["status"] == 0).sum()) == 8
    _run(ctx, api, oracle, batch, band=250, tie_break=1, ins=3, dele=6)


@pytest.mark.parametrize("band,n", [(100, 8), (250, 8)])
def test_c5_50kb_full_length(ctx, api, oracle, synth, band, n):
    """The ultra-long configuration at its own size: 50 kb reads at 20 % error, band 100 and band 250,
    every pair through the wave-per-pair kernels, bit for bit against the oracle (whose matrices take
    ~125 MB per pair at band 250)."""
    batch, cfg = synth.preset("c5_50kb_wide", n)
    pairs, _ = _run(ctx, api, oracle, batch, band=band)
    assert int((pairs["status"] == 0).sum()) == n
    assert int(pairs["rows"].min()) > 45_000 and int(pairs["ncells"].min()) > 45_000 * (2 * band)
    assert ctx.timing()["wide_pairs"] == n


def test_c4_reference_windows_chr20(ctx, api, oracle, synth):
    """The chr20-sized configuration the way it runs: a 64 Mb reference with N runs resident on the
    device, ten candidates per read (the true locus and nine decoys that view the same read), targets
    given as (start, length, strand) windows, about half of them on the reverse strand."""
    n = 400
    ref = synth.reference(synth.CHR20_LEN, n_frac=0.01)
    batch = synth.generate(n, read_len=10_000, seed=4000, decoy_every=10, share_reads=True, ref=ref)
    assert 100 < int(batch["win_rc"].sum()) < 300 and len(np.unique(batch["read_id"])) == 40
    ctx.set_scoring(api.make_scoring())
    ctx.set_reference(ref)
    try:
        pairs, blocks, rc = ctx.refine_host(synth.as_windows(batch))
    finally:
        ctx.set_reference(None)
    assert rc == 0
    util.compare_batch(oracle, batch, oracle.make_params(), pairs, blocks)     # the oracle aligns the spelled-out targets
    assert int((pairs["status"] == 0).sum()) >= n - 5
    # a window that starts inside an N run of the reference is handled like any other
    touches_n = [p for p in range(n) if (synth.pair(batch, p)[1] == ord("N")).any()]
    assert all(int(pairs["status"][p]) in (0, 3) for p in touches_n)


def test_recorded_fuzz_mismatches(ctx, api, oracle, synth):
    """Every pair on which a randomised run once disagreed with the oracle, replayed alone and at
    the batch position it had (second pair, between unrelated pairs), several times."""
    cases = json.load(open(os.path.join(GOLD, "fuzz_regressions.json")))["cases"]
    fill = synth.generate(6, read_len=300, seed=12)
    fillers = [tuple(synth.pair(fill, i)) for i in range(6)]
    fillers = [(bytes(q), bytes(t), [tuple(int(v) for v in b) for b in blk]) for q, t, blk in fillers]
    for c in cases:
        kw = dict(c["scoring"])
        me = (c["q"], c["t"], [tuple(b) for b in c["blocks"]])
        for lst in ([me], fillers[:1] + [me] + fillers[1:]):
            batch = util.make_batch(lst)
            for _ in range(4):
                pairs, _ = _run(ctx, api, oracle, batch, **kw)
            assert int(pairs[1 if len(lst) > 1 else 0]["status"]) == 0, c["name"]


def test_affine_wide_bands(ctx, api, oracle, synth):
    """Affine rows wider than the lane-per-pair classes: the wave-per-pair affine kernel."""
    from blasr_amd import _lib
    batch = synth.generate(40, read_len=2500, len_jitter=1200, p_ins=0.11, p_del=0.06, p_sub=0.03, seed=5003)
    for kw in (dict(band=70), dict(band=130, affine_open=4, affine_extend=1), dict(band=250, ins=7, dele=2, tie_break=1),
               dict(band=90, affine_open=0, affine_extend=0), dict(band=100, dele=20, affine_open=2, affine_extend=9)):
        pairs, _ = _run(ctx, api, oracle, batch, mode=_lib.MODE_AFFINE, **kw)
        assert int((pairs["status"] == 0).sum()) == 40


def test_wide_mixed_classes(ctx, api, oracle, synth):
    # one batch whose pairs spread over the lane-per-pair classes and the wave-per-pair class
    batch = synth.generate(160, read_len=2500, len_jitter=1500, p_ins=0.12, p_del=0.08, p_sub=0.03, seed=5002)
    _run(ctx, api, oracle, batch, band=30)
    _run(ctx, api, oracle, batch, band=55)


@pytest.mark.parametrize("kw", [dict(tie_break=1), dict(n_score=3), dict(band=0), dict(band=3),
                                dict(ins=2, dele=7, indel=3), dict(match=-1, mismatch=1, ins=1, dele=1, indel=1),
                                dict(stats_affine=1, stats_affine_open=4, stats_affine_extend=1),
                                dict(max_band=12)])
def test_parameter_variants(ctx, api, oracle, synth, kw):
    batch = synth.generate(64, read_len=1500, len_jitter=800, seed=31)
    _run(ctx, api, oracle, batch, **kw)


@pytest.mark.parametrize("kw", [dict(), dict(ins=100, dele=100), dict(affine_open=4, affine_extend=1),
                                dict(affine_open=0, affine_extend=0), dict(tie_break=1, affine_open=3, affine_extend=2),
                                dict(ins=100, dele=100, affine_open=6, affine_extend=1, stats_affine=1,
                                     stats_affine_open=6, stats_affine_extend=1)])
def test_affine_variants(ctx, api, oracle, synth, kw):
    from blasr_amd import _lib
    batch = synth.generate(96, read_len=1500, len_jitter=800, seed=41)
    _run(ctx, api, oracle, batch, mode=_lib.MODE_AFFINE, band=16, **kw)


def test_c3_affine_20kb(ctx, api, oracle, synth):
    from blasr_amd import _lib
    batch, cfg = synth.preset("c3_20kb_affine", 70)
    pairs, _ = _run(ctx, api, oracle, batch, mode=_lib.MODE_AFFINE, band=cfg["band"])
    assert (pairs["status"] == 0).all()


def test_affine_edge_cases(ctx, api, oracle):
    from blasr_amd import _lib
    q = "ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGCTA"
    pairs_list = [
        (q, q, [(0, 0, len(q))]),
        (q, q, []),
        ("ACGTACGTACGT" + "GGG" + "TTGACCATGCAA", "ACGTACGTACGT" + "TTGACCATGCAA", [(0, 0, 12), (15, 12, 12)]),
        ("ACGTACGTACGT" + "TTGACCATGCAA", "ACGTACGTACGT" + "CCCCC" + "TTGACCATGCAA", [(0, 0, 12), (12, 17, 12)]),
        ("A", "A", [(0, 0, 1)]),
        ("ACGTNNNNACGTACGT", "ACGTACGTACGTACGT", [(0, 0, 4), (8, 8, 8)]),
    ]
    batch = util.make_batch(pairs_list)
    _run(ctx, api, oracle, batch, expect_status=[0, 1, 0, 0, 0, 0], mode=_lib.MODE_AFFINE, band=16,
         ins=100, dele=100, affine_open=10, affine_extend=1)


def test_custom_matrix(ctx, api, oracle, synth):
    rng = np.random.default_rng(7)
    sub = rng.integers(-9, 12, size=(5, 5))
    np.fill_diagonal(sub, -6)
    batch = synth.generate(48, read_len=1200, seed=32)
    _run(ctx, api, oracle, batch, sub=sub.tolist())


def test_edge_cases_in_one_batch(ctx, api, oracle):
    q = "ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGCTA"
    pairs_list = [
        (q, q, [(0, 0, len(q))]),                               # perfect
        (q, q, []),                                             # empty guide: passed through
        (q, q, [(0, 0, len(q) + 1)]),                           # block past the end
        (q, q, [(0, 0, 10), (5, 12, 5)]),                       # overlapping blocks
        (q.lower(), q, [(0, 0, 12), (20, 20, 10)]),             # lower case query
        ("ACGTNNNNACGTACGT", "ACGTACGTACGTACGT", [(0, 0, 4), (8, 8, 8)]),   # N bases
        ("A", "A", [(0, 0, 1)]),                                # smallest possible pair
        (q, q + "TTTTTTTT", [(3, 5, 11)]),                      # guide not starting at 0, target longer
        ("ACGTACGTAC" + "GGCCTTAACC", "ACGTACGTAC" + "T" * 30 + "GGCCTTAACC", [(0, 0, 10), (10, 40, 10)]),
        ("ACGTACGTAC" + "G" * 25 + "GGCCTTAACC", "ACGTACGTAC" + "GGCCTTAACC", [(0, 0, 10), (35, 10, 10)]),
    ]
    batch = util.make_batch(pairs_list)
    expect = [0, 1, 2, 2, 0, 0, 0, 0, 0, 0]
    pairs, blocks = _run(ctx, api, oracle, batch, expect_status=expect)
    assert int(pairs[1]["n_blk"]) == 0 and int(pairs[7]["q_pos"]) == 3 and int(pairs[7]["t_pos"]) == 5


def test_shared_query_arena(ctx, api, oracle, synth):
    """All candidates of a read view the same query bytes (BlasrUtilsImpl.hpp:700-703)."""
    base = synth.generate(6, read_len=800, seed=50)
    n = 6
    q_off = np.repeat(base["q_off"][:1], n)             # every pair views read 0 ...
    q_len = np.repeat(base["q_len"][:1], n)
    b = dict(base)
    b["q_off"], b["q_len"] = q_off, q_len
    # ... against its own window, with a sparse guide that fits both
    blocks, blk_off, n_blk = [], [], []
    for i in range(n):
        L = int(min(q_len[i], base["t_len"][i]))
        blk_off.append(len(blocks)); n_blk.append(2)
        blocks += [(0, 0, 11), (L - 11, L - 11, 11)]
    arr = np.zeros(len(blocks), dtype=synth.BLOCK_DT)
    for i, x in enumerate(blocks):
        arr[i] = x
    b["blocks"], b["blk_off"], b["n_blk"] = arr, np.array(blk_off, np.uint64), np.array(n_blk, np.uint32)
    _run(ctx, api, oracle, b, band=30)


def test_kband_global(ctx, api, oracle, synth):
    base = synth.generate(40, read_len=700, len_jitter=300, seed=60)
    base["n_blk"][:] = 0                                   # k-band ignores the guide
    from blasr_amd import _lib
    _run(ctx, api, oracle, base, mode=_lib.MODE_KBAND, kband_k=60)
    pairs, _ = _run(ctx, api, oracle, base, mode=_lib.MODE_KBAND, kband_k=8)
    assert (pairs["status"] == _lib.PAIR_END_OUT_OF_BAND).any()


def test_device_path_equals_host_path(ctx, api, oracle, synth):
    import torch
    from blasr_amd import _lib
    batch = synth.generate(256, read_len=2000, len_jitter=1000, seed=70)
    ctx.set_scoring(api.make_scoring())
    hp, hb, rc = ctx.refine_host(batch)
    d = api.to_device(batch)
    n = 256
    out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
    cap = int(ctx.blocks_needed)
    out_blocks = torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0")
    rc, needed = ctx.refine_device(d, out_pairs, out_blocks)
    assert rc == 0 and needed == cap
    dp = out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT)
    db = out_blocks.cpu().numpy().view(np.uint32)
    assert np.array_equal(dp, hp)
    assert np.array_equal(db[:, 0], hb["q_pos"]) and np.array_equal(db[:, 1], hb["t_pos"]) and np.array_equal(db[:, 2], hb["length"])
    tm = ctx.timing()
    assert tm["cells"] == int(hp["ncells"].sum()) and tm["fill_ms"] > 0
    # too small an arena: GBA_E_NOSPACE, the needed size is reported, leading pairs still fit
    small = torch.zeros((cap // 2, 3), dtype=torch.int32, device="cuda:0")
    rc, needed = ctx.refine_device(d, out_pairs, small)
    assert rc == _lib.E_NOSPACE and needed == cap
    dp2 = out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT)
    assert (dp2["status"] == _lib.PAIR_NOSPACE).any() and dp2["status"][0] == 0


def test_host_nospace_and_reuse(ctx, api, oracle, synth):
    from blasr_amd import _lib
    ctx.set_scoring(api.make_scoring())
    big = synth.generate(128, read_len=3000, seed=80)
    small = synth.generate(16, read_len=500, seed=81)
    p1, b1, rc = ctx.refine_host(big)
    p2, b2, rc = ctx.refine_host(small)                  # scratch is reused by a smaller batch
    util.compare_batch(oracle, small, oracle.make_params(), p2, b2)
    p3, b3, rc = ctx.refine_host(big, blocks_capacity=100)
    assert rc == _lib.E_NOSPACE and ctx.blocks_needed == len(b1)
    assert (p3["status"] == _lib.PAIR_NOSPACE).sum() > 100


def test_unsupported_scoring_is_refused(ctx, api, oracle, synth):
    """A refused scoring leaves the previous one in force, whichever check refuses it."""
    from blasr_amd import _lib
    batch = synth.generate(40, read_len=400, seed=31)
    kw = dict(match=-4, mismatch=7, ins=6, dele=3)
    ctx.set_scoring(api.make_scoring(**kw))
    for bad in (dict(mismatch=500), dict(mode=_lib.MODE_AFFINE, mismatch=100),
                dict(mode=_lib.MODE_AFFINE, affine_open=60, affine_extend=10)):
        with pytest.raises(_lib.GbaError) as e:
            ctx.set_scoring(api.make_scoring(**bad))
        assert e.value.code == _lib.E_UNSUPPORTED
        pairs, blocks, rc = ctx.refine_host(batch)
        assert rc == 0
        util.compare_batch(oracle, batch, oracle.make_params(**kw), pairs, blocks)
    ctx.set_scoring(api.make_scoring())


def test_incomplete_batch_is_refused(ctx, api, synth):
    """Missing arrays are an argument error, not a crash; no reference set is a sequence error."""
    import ctypes as C
    from blasr_amd import _lib
    ctx.set_scoring(api.make_scoring())
    batch = synth.generate(4, read_len=100, seed=5)
    pairs = np.zeros(4, dtype=_lib.PAIR_RESULT_DT)
    blocks = np.zeros(500, dtype=_lib.BLOCK_DT)
    r = _lib.Results(pairs.ctypes.data, blocks.ctypes.data, 500, 0)
    b, keep = api.make_batch_struct(batch)
    b.q_len = None
    assert _lib.lib().gba_submit(ctx._h, C.byref(b), C.byref(r)) == _lib.E_INVALID
    b, keep = api.make_batch_struct(batch)
    b.t_arena = None
    assert _lib.lib().gba_submit(ctx._h, C.byref(b), C.byref(r)) == _lib.E_STATE
    assert _lib.lib().gba_wait(ctx._h) == _lib.E_STATE       # nothing was submitted
    pairs, blocks, rc = ctx.refine_host(batch)          # the ctx is still usable
    assert rc == 0 and (pairs["status"] == 0).all()


def test_slices_outside_their_arenas(ctx, api, oracle, synth):
    """Offsets or lengths that leave an arena give that pair BAD_GUIDE; nothing is read there and
    the other pairs are unaffected."""
    from blasr_amd import _lib
    batch = synth.generate(64, read_len=300, seed=9)
    ctx.set_scoring(api.make_scoring())
    good, gblocks, rc = ctx.refine_host(batch)
    bad = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
    bad["q_off"][3] = np.uint64(1) << np.uint64(40)
    bad["t_off"][7] = np.uint64(len(batch["t_arena"]) - 5)
    bad["blk_off"][11] = np.uint64(len(batch["blocks"]))
    bad["q_len"][13] = np.uint32(0xfffffff0)
    bad["t_off"][17] = np.uint64(0xffffffffffffff00)
    pairs, blocks, rc = ctx.refine_host(bad, blocks_capacity=len(gblocks))
    assert rc == 0
    hit = [3, 7, 11, 13, 17]
    assert (pairs["status"][hit] == _lib.PAIR_BAD_GUIDE).all()
    rest = np.setdiff1d(np.arange(64), hit)
    for f in util.FIELDS:
        assert np.array_equal(pairs[f][rest], good[f][rest]), f
    for p in rest:
        a = blocks[int(pairs["blk_off"][p]):int(pairs["blk_off"][p]) + int(pairs["n_blk"][p])]
        b = gblocks[int(good["blk_off"][p]):int(good["blk_off"][p]) + int(good["n_blk"][p])]
        assert np.array_equal(a, b)
    for mode_kw in (dict(mode=_lib.MODE_KBAND, kband_k=40),):
        ctx.set_scoring(api.make_scoring(**mode_kw))
        pairs, blocks, rc = ctx.refine_host(bad, blocks_capacity=int(batch["q_len"].sum()) + 64)
        assert rc == 0 and (pairs["status"][[3, 7, 13, 17]] != _lib.PAIR_OK).all()
    ctx.set_scoring(api.make_scoring())


def test_many_short_pairs(ctx, api, oracle, synth):
    """150 000 pairs in one batch: the single-block scans and the class counters over many iterations."""
    from concurrent.futures import ThreadPoolExecutor
    n = 150_000
    batch = synth.generate(n, read_len=120, len_jitter=60, seed=4, ref_len=500_000, min_block=5,
                           p_ins=0.03, p_del=0.02, p_sub=0.01)
    ctx.set_scoring(api.make_scoring())
    pairs, blocks, rc = ctx.refine_host(batch)
    assert rc == 0
    P = oracle.make_params()
    chunk = n // 8
    with ThreadPoolExecutor(8) as pool:
        parts = list(pool.map(lambda f: oracle.align_batch(batch, P, f, min(chunk, n - f)), range(0, n, chunk)))
    ores = np.concatenate([r for r, _ in parts])
    assert np.array_equal(ores["status"], pairs["status"])
    for f in util.FIELDS:
        assert np.array_equal(ores[f].astype(np.int64), pairs[f].astype(np.int64)), f
    ob = np.concatenate([x for _, bl in parts for x in bl if len(x)])
    assert len(ob) == len(blocks)
    for f in ("q_pos", "t_pos", "length"):
        assert np.array_equal(ob[f], blocks[f]), f
    # the same batch again and again: results that depend on timing (a late memory operation, a race
    # between the classes' launches) show up as a difference in some of the runs
    for _ in range(12):
        again, ablocks, rc = ctx.refine_host(batch)
        assert rc == 0
        for f in util.FIELDS:
            assert np.array_equal(again[f], pairs[f]), f
        assert np.array_equal(ablocks, blocks)


def test_sparse_class_in_a_large_batch(ctx, api, oracle, synth):
    """More than 5 GB of traceback records with a class that holds one pair in 680: the 64 pairs of
    that class's only wave lie across the whole batch, their records must not."""
    n = 44_000
    batch = synth.generate(n, read_len=10_000, seed=23, drop_frac=0.0, p_ins=0.03, p_del=0.02, p_sub=0.01)
    batch = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
    rare = np.arange(137, n, 680)
    for p in rare:                                   # keep every 60th block: long gaps, rows of 60 to 120 cells
        bo, nb = int(batch["blk_off"][p]), int(batch["n_blk"][p])
        keep = np.r_[np.arange(0, nb - 1, 60), nb - 1]
        batch["blocks"][bo:bo + len(keep)] = batch["blocks"][bo:bo + nb][keep]
        batch["n_blk"][p] = len(keep)
    kw = dict()                                      # at these error rates everything else fits the 64-slot class
    ctx.set_scoring(api.make_scoring(**kw))
    pairs, blocks, rc = ctx.refine_host(batch)
    assert rc == 0
    P = oracle.make_params(**kw)
    rng = np.random.default_rng(1)
    for p in np.r_[rare, rng.integers(0, n, 60)]:
        q, t, b = synth.pair(batch, int(p))
        o, ob = oracle.align(q, t, b, P)
        k = pairs[p]
        assert int(k["status"]) == o["status"], (p, int(k["status"]), o["status"])
        if o["status"] != 0:
            continue
        for f in util.FIELDS:
            assert int(o[f]) == int(k[f]), (f, p)
        kb = blocks[int(k["blk_off"]):int(k["blk_off"]) + int(k["n_blk"])]
        assert np.array_equal(kb, ob), p
    assert int((pairs["status"] == 0).sum()) >= n - len(rare)


def test_full_size_properties(ctx, api, oracle, synth):
    """Benchmark-sized pairs: size-independent invariants on all pairs, oracle parity on a sample."""
    batch, cfg = synth.preset("c2_10kb_linear", 2048)
    ctx.set_scoring(api.make_scoring())
    pairs, blocks, rc = ctx.refine_host(batch)
    assert rc == 0 and (pairs["status"] == 0).all()
    assert np.array_equal(pairs["n_match"] + pairs["n_mismatch"] + pairs["n_ins"], pairs["rows"])
    assert np.array_equal(pairs["n_match"] + pairs["n_mismatch"] + pairs["n_del"], pairs["cols"])
    assert np.array_equal(pairs["rows"], batch["q_len"]) and np.array_equal(pairs["cols"], batch["t_len"])
    off = pairs["blk_off"].astype(np.int64)
    assert np.array_equal(off[1:], np.cumsum(pairs["n_blk"].astype(np.int64))[:-1])
    # blocks of every pair: increasing, inside the pair, lengths add up to the aligned columns
    for p in range(0, 2048, 37):
        b = blocks[off[p]:off[p] + int(pairs["n_blk"][p])]
        qe = b["q_pos"].astype(np.int64) + b["length"]
        te = b["t_pos"].astype(np.int64) + b["length"]
        assert np.all(b["q_pos"][1:] >= qe[:-1]) and np.all(b["t_pos"][1:] >= te[:-1])
        assert int(b["length"].sum()) == int(pairs["n_match"][p] + pairs["n_mismatch"][p])
        # rescoring the path on the host reproduces the reported score
        q, t, _ = synth.pair(batch, p)
        sc = 0
        for x in b:
            qs = q[x["q_pos"]:x["q_pos"] + x["length"]]
            ts = t[x["t_pos"]:x["t_pos"] + x["length"]]
            eq = int((qs == ts).sum())
            sc += -5 * eq + 6 * (len(qs) - eq)
        sc += 5 * int(pairs["n_ins"][p] + pairs["n_del"][p])
        assert sc == int(pairs["score"][p])
    sample = {k: v for k, v in batch.items()}
    ores, oblk = oracle.align_batch(sample, oracle.make_params(), 0, 24)
    for p in range(24):
        for f in util.FIELDS:
            assert int(ores[p][f]) == int(pairs[p][f]), (f, p)


_COMP = np.full(256, ord("N"), dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[_a] = _b


def test_reference_windows_and_reverse_complement(ctx, api, oracle, synth):
    """Targets given as windows of a reference resident in device memory, half of them on the
    reverse strand (the reference's genome.Copy / CopyAsRC, iblasr/BlasrAlignImpl.hpp:727-743):
    same results as the same targets spelled out in an arena, which the oracle checks."""
    from blasr_amd import _lib
    n = 96
    batch = synth.generate(n, read_len=1800, len_jitter=900, seed=77)
    # build a "genome": the pairs' targets laid end to end with spacers, odd pairs stored as the
    # reverse complement of what the aligner must see
    pieces, t_off, t_rc, pos = [], np.zeros(n, np.uint64), np.zeros(n, np.uint8), 0
    for p in range(n):
        t = np.asarray(batch["t_arena"][int(batch["t_off"][p]):int(batch["t_off"][p]) + int(batch["t_len"][p])])
        if p % 2:
            t = _COMP[t[::-1]]
            t_rc[p] = 1
        spacer = np.frombuffer(b"ACGTN" * 3, dtype=np.uint8)
        pieces += [spacer, t]
        pos += spacer.size
        t_off[p] = pos
        pos += t.size
    genome = np.concatenate(pieces)
    ctx.set_scoring(api.make_scoring())
    want, wblk, rc = ctx.refine_host(batch)
    assert rc == 0
    util.compare_batch(oracle, batch, oracle.make_params(), want, wblk, None)
    ctx.set_reference(genome)
    wb = {k: batch[k] for k in ("n_pairs", "q_arena", "q_off", "q_len", "t_len", "blocks", "blk_off", "n_blk")}
    wb["t_off"] = t_off
    wb["t_rc"] = t_rc
    got, gblk, rc = ctx.refine_host(wb)
    assert rc == 0
    assert np.array_equal(got, want) and np.array_equal(gblk, wblk)
    # a window that leaves the reference is refused for that pair only
    bad = dict(wb)
    bad["t_off"] = t_off.copy()
    bad["t_off"][5] = genome.size - 10
    got2, _, rc = ctx.refine_host(bad)
    assert rc == 0 and int(got2[5]["status"]) == _lib.PAIR_BAD_GUIDE and int(got2[6]["status"]) == 0
    # device-resident batch against the same reference
    import torch
    d = api.to_device(wb)
    out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
    out_blocks = torch.zeros((len(wblk) + 8, 3), dtype=torch.int32, device="cuda:0")
    rc, needed = ctx.refine_device(d, out_pairs, out_blocks)
    assert rc == 0 and needed == len(wblk)
    assert np.array_equal(out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT), want)
    ctx.set_reference(None)
    with pytest.raises(Exception):
        ctx.refine_host(wb)          # no reference, no arena: call out of sequence


@pytest.mark.parametrize("part", ["64", "128", "320"])
def test_batches_in_parts(api, oracle, synth, monkeypatch, part):
    """GBA_PART_PAIRS: a batch run as 6, 3 or 2 staggered parts over the two slots gives the same
    results and the same packed block arena as in one piece (a small arena included)."""
    monkeypatch.setenv("GBA_PART_PAIRS", part)
    c2 = api.Context(0)
    try:
        batch = synth.generate(333, read_len=1200, len_jitter=900, p_ins=0.1, p_del=0.07, seed=88, decoy_every=7)
        for kw in (dict(), dict(band=60), dict(mode=1, band=16)):
            _run(c2, api, oracle, batch, **kw)
        c2.set_scoring(api.make_scoring())
        full, fblocks, rc = c2.refine_host(batch)
        assert rc == 0
        small, sblocks, rc = c2.refine_host(batch, blocks_capacity=len(fblocks) // 2)
        assert rc == -4 and c2.blocks_needed == len(fblocks)
        fits = full["blk_off"].astype(np.int64) + full["n_blk"] <= len(fblocks) // 2
        ok = full["status"] == 0
        assert (small["status"][ok & ~fits] == 4).all() and (small["status"][ok & fits] == 0).all()
        assert np.array_equal(small["blk_off"], full["blk_off"])
        end = int((full["blk_off"].astype(np.int64) + full["n_blk"])[ok & fits].max())
        assert np.array_equal(sblocks[:end], fblocks[:end])
    finally:
        c2.close()


@pytest.mark.parametrize("wave_pairs,n", [("100", 400), ("100", 250), ("300", 400)])
def test_host_batch_of_one_part_runs_as_two_halves(api, oracle, synth, monkeypatch, wave_pairs, n):
    """A host-buffer batch of 2/5 of a part and more, which would be one part, runs as two halves when the halves still
    reach the lane classes (400 and 250 pairs of parts of 600 with lanes from 100 pairs); otherwise in one piece (lanes
    from 300): results and the packed block arena are the oracle's either way, and a small arena fails the same pairs."""
    monkeypatch.setenv("GBA_PART_PAIRS", "600")
    monkeypatch.setenv("GBA_WAVE_PAIRS", wave_pairs)
    monkeypatch.setenv("GBA_WAVE_PAIRS_AFFINE", wave_pairs)
    c2 = api.Context(0, timing=True)
    try:
        batch = synth.generate(n, read_len=1200, len_jitter=900, p_ins=0.1, p_del=0.07, seed=89, decoy_every=7)
        launches = []
        for kw in (dict(), dict(mode=1, band=16)):
            _run(c2, api, oracle, batch, **kw)
            launches.append(c2.timing()["fill_launches"])
        monkeypatch.setenv("GBA_PART_PAIRS", "0")
        c1 = api.Context(0, timing=True)
        try:
            for i, kw in enumerate((dict(), dict(mode=1, band=16))):
                _run(c1, api, oracle, batch, **kw)
                whole = c1.timing()["fill_launches"]              # the batch in one piece: one launch per class
                assert launches[i] > whole if wave_pairs == "100" else launches[i] == whole
        finally:
            c1.close()
        c2.set_scoring(api.make_scoring())
        full, fblocks, rc = c2.refine_host(batch)
        assert rc == 0
        small, sblocks, rc = c2.refine_host(batch, blocks_capacity=len(fblocks) // 2)
        assert rc == -4 and c2.blocks_needed == len(fblocks)
        fits = full["blk_off"].astype(np.int64) + full["n_blk"] <= len(fblocks) // 2
        ok = full["status"] == 0
        assert (small["status"][ok & ~fits] == 4).all() and (small["status"][ok & fits] == 0).all()
    finally:
        c2.close()


@pytest.mark.parametrize("edge,part", [("1", "128"), ("1", "600"), ("1", "400"), ("0", "128")])
def test_device_batches_with_small_first_and_last_parts(api, synth, monkeypatch, edge, part):
    """A context that has the GPU to itself cuts a device-resident batch into parts of its own shape (two halves, or
    3 : 4 : ... : 4 : 3; GBA_EDGE_PARTS=0: equal parts): pair results and the packed block arena are those of the
    host-buffer path, whose parts are equal; a second context with a batch in flight at the same time gets equal
    parts, and the same bytes."""
    import threading
    import torch
    from blasr_amd import _lib
    monkeypatch.setenv("GBA_PART_PAIRS", part)
    monkeypatch.setenv("GBA_EDGE_PARTS", edge)
    n = 900
    batch = synth.generate(n, read_len=1500, len_jitter=900, seed=606, decoy_every=6)
    ctxs = [api.Context(0), api.Context(0)]
    try:
        for c in ctxs:
            c.set_scoring(api.make_scoring())
        want, wblk, rc = ctxs[0].refine_host(batch)
        assert rc == 0
        d = api.to_device(batch)
        outs = [(torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0"),
                 torch.zeros((len(wblk) + 8, 3), dtype=torch.int32, device="cuda:0")) for _ in ctxs]
        rc, needed = ctxs[0].refine_device(d, *outs[0])
        torch.cuda.synchronize()
        assert rc == 0 and needed == len(wblk)
        assert outs[0][0].cpu().numpy().tobytes() == want.tobytes()
        assert outs[0][1][:needed].cpu().numpy().tobytes() == wblk.tobytes()
        for o in outs:
            o[0].zero_(); o[1].zero_()
        got = [None, None]

        def run(i):
            for _ in range(3):
                got[i] = ctxs[i].refine_device(d, *outs[i])
        th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        for i in range(2):
            assert got[i] == (0, len(wblk))
            assert outs[i][0].cpu().numpy().tobytes() == want.tobytes()
            assert outs[i][1][:len(wblk)].cpu().numpy().tobytes() == wblk.tobytes()
    finally:
        for c in ctxs:
            c.close()


def test_host_batches_pinned_async_and_out_of_order(api, oracle, synth, monkeypatch):
    """gba_submit returns before the batch has run; batches and results in page-locked memory of
    the ctx; arenas whose pairs are NOT laid out in pair order (each part of the batch then needs
    stretches on both sides of what is on the device already); all in seven staggered parts."""
    monkeypatch.setenv("GBA_PART_PAIRS", "128")
    c2 = api.Context(0)
    try:
        n = 900
        batch = synth.generate(n, read_len=1500, len_jitter=900, seed=505, decoy_every=6)
        rng = np.random.default_rng(3)
        perm = rng.permutation(n)
        shuffled = dict(batch)
        for k in ("q_off", "q_len", "t_off", "t_len", "blk_off", "n_blk"):
            shuffled[k] = batch[k][perm].copy()
        c2.set_scoring(api.make_scoring())
        want, wblk, rc = c2.refine_host(batch)
        assert rc == 0
        util.compare_batch(oracle, batch, oracle.make_params(), want, wblk)
        pb = c2.pinned_copy(shuffled)
        pairs = c2.host_array(n, _lib_dt().PAIR_RESULT_DT)
        blocks = c2.host_array(len(wblk) + 5, _lib_dt().BLOCK_DT)
        for rep in range(3):
            pairs[...] = 0
            c2.submit_host(pb, pairs, blocks)
            with pytest.raises(Exception):                 # one batch in flight per ctx
                c2.submit_host(pb, pairs, blocks)
            rc, needed = c2.wait_host()
            assert rc == 0 and needed == len(wblk)
            for f in util.FIELDS:
                assert np.array_equal(pairs[f], want[f][perm]), f
            for p in rng.integers(0, n, 40):
                a = blocks[int(pairs["blk_off"][p]):int(pairs["blk_off"][p]) + int(pairs["n_blk"][p])]
                o = perm[p]
                b = wblk[int(want["blk_off"][o]):int(want["blk_off"][o]) + int(want["n_blk"][o])]
                assert np.array_equal(a, b)
        # an arena too small by half: the parts that fit are delivered, the rest is NOSPACE
        small = c2.host_array(len(wblk) // 2, _lib_dt().BLOCK_DT)
        c2.submit_host(pb, pairs, small)
        rc, needed = c2.wait_host()
        assert rc == -4 and needed == len(wblk)
        fits = pairs["blk_off"].astype(np.int64) + pairs["n_blk"] <= len(small)
        ok = want["status"][perm] == 0
        assert (pairs["status"][ok & ~fits] == 4).all() and (pairs["status"][ok & fits] == 0).all()
        end = int((pairs["blk_off"].astype(np.int64) + pairs["n_blk"])[ok & fits].max())
        c2.submit_host(pb, pairs, blocks)
        c2.wait_host()
        assert np.array_equal(small[:end], blocks[:end])
    finally:
        c2.close()


def _lib_dt():
    from blasr_amd import _lib
    return _lib


def test_extra_rescoring_scorers(ctx, api, oracle, synth):
    """The path rescored with further scorers (the reference's probability score and edit distance,
    iblasr/BlasrUtilsImpl.hpp:120-151, :1056-1063).  One extra scorer repeats the stats scorer with
    affine gap runs, so it must equal `score` (which the oracle checks); the other is a random
    matrix with linear gaps, recomputed here from the blocks; a third run checks the edit distance."""
    from blasr_amd import _lib
    rng = np.random.default_rng(5)
    batch = synth.generate(200, read_len=1500, len_jitter=800, seed=91, decoy_every=9)
    code = np.full(256, 4, np.int64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i; code[ch + 32] = i
    for mode, band in ((_lib.MODE_LINEAR, 10), (_lib.MODE_LINEAR, 90), (_lib.MODE_AFFINE, 16)):
        kw = dict(mode=mode, band=band, stats_affine=1, stats_affine_open=6, stats_affine_extend=2)
        dp = [[(0 if a == 4 or b == 4 else -5 if a == b else 6) for b in range(5)] for a in range(5)]
        m1 = rng.integers(-40, 40, size=(5, 5))
        extra = [dict(sub=m1.tolist(), ins=3, dele=8), dict(sub=dp, affine=1, affine_open=6, affine_extend=2)]
        ctx.set_scoring(api.make_scoring(extra=extra, **kw))
        got, gb, rc = ctx.refine_host(batch)
        assert rc == 0
        util.compare_batch(oracle, batch, oracle.make_params(**kw), got, gb, None)
        ok = got["status"] == 0
        assert ok.sum() > 100
        assert np.array_equal(got["extra_score"][ok, 1], got["score"][ok])
        for p in np.flatnonzero(ok)[:80]:
            k = got[p]
            blk = gb[int(k["blk_off"]):int(k["blk_off"]) + int(k["n_blk"])]
            q0 = int(batch["q_off"][p]) + int(k["q_pos"]); t0 = int(batch["t_off"][p]) + int(k["t_pos"])
            sc = 3 * int(k["n_ins"]) + 8 * int(k["n_del"])
            for b in blk:
                bq, bt, ln = int(b["q_pos"]), int(b["t_pos"]), int(b["length"])
                qs = code[batch["q_arena"][q0 + bq:q0 + bq + ln]]; ts = code[batch["t_arena"][t0 + bt:t0 + bt + ln]]
                sc += int(m1[qs, ts].sum())
            assert sc == int(k["extra_score"][0]), (p, sc, int(k["extra_score"][0]))
    # edit distance: mismatches + gap bases
    edit = [[0 if a == b else 1 for b in range(5)] for a in range(5)]
    ctx.set_scoring(api.make_scoring(extra=[dict(sub=edit, ins=1, dele=1)]))
    got, gb, rc = ctx.refine_host(batch)
    ok = got["status"] == 0
    nm = got["n_mismatch"].astype(np.int64) + got["n_ins"] + got["n_del"]
    assert np.array_equal(got["extra_score"][ok, 0], nm[ok]) and not got["extra_score"][ok, 1].any()


def _pair_with_rows(synth, rows, seed):
    """A synthetic pair cut so that its DP region has exactly `rows` query bases: the guide's
    prefix, closed by an 11-base block on the diagonal of the last kept block."""
    batch = synth.generate(1, read_len=rows + 600, seed=seed)
    q, t, blk = synth.pair(batch, 0)
    keep = [b for b in blk if int(b["q_pos"]) + int(b["length"]) <= rows - 40]
    lq = int(keep[-1]["q_pos"]) + int(keep[-1]["length"])
    lt = int(keep[-1]["t_pos"]) + int(keep[-1]["length"])
    keep = [tuple(int(v) for v in b) for b in keep] + [(rows - 11, lt + (rows - 11 - lq), 11)]
    tl = keep[-1][1] + 11
    assert tl <= len(t) and keep[0][0] == 0
    return bytes(q[:rows]), bytes(t[:tl]), keep


@pytest.mark.parametrize("mode,limit", [("linear", 65535), ("affine", 32767)])
def test_maximum_rows(ctx, api, oracle, synth, mode, limit):
    """The longest DP region the kernels take (65 535 rows, affine 32 767) is bit-exact with the
    oracle; one row more is handed back untouched with a status of its own."""
    from blasr_amd import _lib
    kw = dict(mode=_lib.MODE_AFFINE, band=16) if mode == "affine" else {}
    ok = _pair_with_rows(synth, limit, 31)
    over = _pair_with_rows(synth, limit + 1, 32)
    batch = util.make_batch([ok, over, ok])
    ctx.set_scoring(api.make_scoring(**kw))
    pairs, blocks, rc = ctx.refine_host(batch)
    assert rc == 0
    assert [int(s) for s in pairs["status"]] == [_lib.PAIR_OK, _lib.PAIR_TOO_LONG, _lib.PAIR_OK]
    assert int(pairs["rows"][0]) == limit and int(pairs["n_blk"][1]) == 0
    res, blks = oracle.align(ok[0], ok[1], ok[2], oracle.make_params(**kw))
    for p in (0, 2):
        for f in util.FIELDS:
            assert int(res[f]) == int(pairs[p][f]), (f, p)
        kb = blocks[int(pairs[p]["blk_off"]):int(pairs[p]["blk_off"]) + int(pairs[p]["n_blk"])]
        assert np.array_equal(kb["q_pos"], blks["q_pos"]) and np.array_equal(kb["t_pos"], blks["t_pos"])
        assert np.array_equal(kb["length"], blks["length"])


def test_score_range_of_the_cell_words(ctx, api, oracle, synth):
    """At the largest costs the kernels accept, a pair is aligned (bit-exact) while its worst-case
    score fits the cell words, and handed back TOO_LONG, not misreported, once it does not:
    rows x (128 + 127) + columns x 127 against 2^24."""
    from blasr_amd import _lib
    kw = dict(mismatch=127, ins=127, dele=127, indel=9)
    fits = _pair_with_rows(synth, 42_000, 33)            # 42 000 x 255 + ~40 600 x 127 < 2^24
    over = _pair_with_rows(synth, 46_000, 34)
    batch = util.make_batch([fits, over])
    ctx.set_scoring(api.make_scoring(**kw))
    pairs, blocks, rc = ctx.refine_host(batch)
    assert rc == 0 and [int(x) for x in pairs["status"]] == [_lib.PAIR_OK, _lib.PAIR_TOO_LONG]
    res, blks = oracle.align(fits[0], fits[1], fits[2], oracle.make_params(**kw))
    for f in util.FIELDS:
        assert int(res[f]) == int(pairs[0][f]), f
    kb = blocks[:int(pairs[0]["n_blk"])]
    assert np.array_equal(kb["q_pos"], blks["q_pos"]) and np.array_equal(kb["length"], blks["length"])
    ctx.set_scoring(api.make_scoring())
    pairs, _, rc = ctx.refine_host(batch)                # the same pairs at the default costs both fit
    assert (pairs["status"] == 0).all()


@pytest.mark.parametrize("mode", ["linear", "affine"])
def test_maximum_width(ctx, api, oracle, synth, mode):
    """Rows of 1023 cells (band 511) are the widest the kernels take; band 512 makes every row of
    a pair longer than 1023 cells and the pair comes back TOO_WIDE."""
    from blasr_amd import _lib
    kw = dict(mode=_lib.MODE_AFFINE) if mode == "affine" else {}
    batch = synth.generate(6, read_len=2600, seed=404)
    _run(ctx, api, oracle, batch, band=511, max_band=511, **kw)
    ctx.set_scoring(api.make_scoring(band=512, max_band=512, **kw))
    pairs, blocks, rc = ctx.refine_host(batch)
    assert rc == 0 and (pairs["status"] == _lib.PAIR_TOO_WIDE).all() and (pairs["n_blk"] == 0).all()


def test_four_pairs_per_wave_equals_one_pair_per_wave(api, oracle, synth, monkeypatch):
    """The narrow affine wave class at a batch size where the library itself switches to four pairs per wave (10 243 pairs:
    not a multiple of four, lengths that differ inside a wave): the same bytes as one pair per wave, and the first pairs
    against the oracle."""
    from blasr_amd import _lib
    n = 10243
    batch = synth.compact_blocks(synth.generate(n, read_len=700, len_jitter=500, seed=99))
    kw = dict(mode=_lib.MODE_AFFINE, band=16, affine_open=7, affine_extend=2, stats_affine=1)
    monkeypatch.setenv("GBA_WAVE_PAIRS_AFFINE", "4000000000")
    outs = []
    for quad_min in ("4000000000", None):
        if quad_min is None:
            monkeypatch.delenv("GBA_QUAD_AFFINE_MIN", raising=False)
        else:
            monkeypatch.setenv("GBA_QUAD_AFFINE_MIN", quad_min)
        c = api.Context(0, timing=True)
        c.set_scoring(api.make_scoring(**kw))
        pairs, blocks, rc = c.refine_host(batch)
        assert rc == 0 and int((pairs["status"] == 0).sum()) == n and int(c.timing()["wide_pairs"]) == n
        outs.append((pairs.tobytes(), blocks.tobytes()))
        c.close()
    assert outs[0] == outs[1]
    ores, oblk = oracle.align_batch(batch, oracle.make_params(**kw), 0, 24)
    for p in range(24):
        assert all(int(ores[p][f]) == int(pairs[p][f]) for f in util.FIELDS), p
        kb = blocks[int(pairs[p]["blk_off"]):int(pairs[p]["blk_off"]) + int(pairs[p]["n_blk"])]
        assert kb.tobytes() == np.asarray(oblk[p]).tobytes(), p


@pytest.mark.parametrize("kw", [dict(band=10), dict(band=40, tie_break=1), dict(mode=1, band=16, affine_open=6, affine_extend=1, stats_affine=1)])
def test_small_batches_walk_back_one_wave_per_pair(api, oracle, synth, monkeypatch, kw):
    """A batch small enough to run on the wave-per-pair kernels altogether is also traced back one wave per pair (diagonal
    runs taken lanes side by side); the same bytes as the lane-per-pair traceback gives, and the oracle's results."""
    batch = synth.compact_blocks(synth.generate(300, read_len=900, len_jitter=700, seed=17))
    outs = []
    for tb_max in ("0", None):
        if tb_max is None:
            monkeypatch.delenv("GBA_TB_WAVE_MAX", raising=False)
        else:
            monkeypatch.setenv("GBA_TB_WAVE_MAX", tb_max)
        c = api.Context(0, timing=True)
        pairs, blocks = _run(c, api, oracle, batch, **kw)
        assert int(c.timing()["wide_pairs"]) == 300
        outs.append((pairs.tobytes(), blocks.tobytes()))
        c.close()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("band", [16, 50])
@pytest.mark.parametrize("wave_pairs", ["0", "4000000000"], ids=["lane-classes", "narrow-wave-class"])
def test_affine_lane_traceback_windows_agree(api, oracle, synth, monkeypatch, band, wave_pairs):
    """The lane-per-pair traceback of affine pairs keeps 64 record dwords per lane in its window, or 192 when the launch has
    few pairs (rows of 33 cells are ten dwords; rows of 101 cells are 27 and leave the small window every other row): the
    same bytes either way, and the oracle's results."""
    from blasr_amd import _lib
    batch = synth.compact_blocks(synth.generate(600, read_len=1500, len_jitter=1400, seed=23))
    kw = dict(mode=_lib.MODE_AFFINE, band=band, affine_open=5, affine_extend=2, stats_affine=1)
    monkeypatch.setenv("GBA_WAVE_PAIRS_AFFINE", wave_pairs)
    monkeypatch.setenv("GBA_TB_WAVE_MAX", "0")            # the narrow wave class walked back one lane per pair too
    outs = []
    for few in ("0", None):
        if few is None:
            monkeypatch.delenv("GBA_TB_FEW_MAX", raising=False)
        else:
            monkeypatch.setenv("GBA_TB_FEW_MAX", few)
        c = api.Context(0, timing=True)
        pairs, blocks = _run(c, api, oracle, batch, **kw)
        if wave_pairs != "0":
            assert int(c.timing()["wide_pairs"]) == 600
        outs.append((pairs.tobytes(), blocks.tobytes()))
        c.close()
    assert outs[0] == outs[1]
