"""The tracebacks' block scratch: a pair's own region holds about a block per three rows, a pair with more blocks moves to
a full-size region of the shared pool, and a batch whose pairs exhaust the pool is run again with full-size regions.
Whatever the sizes, the results are byte for byte those of a context with room for the worst case."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ctx_wide_pairs(api, ctx, batch):
    """Pairs of the batch that ran one wave per pair (a context with timing on, same environment)."""
    t = api.Context(0, timing=True)
    try:
        t.set_scoring(ctx.scoring)
        t.refine_host_intervals(batch)
        return t.timing()["wide_pairs"]
    finally:
        t.close()


def _run_all(api, _lib, ctx, batch):
    """Host path with blocks and CIGARs, host path in intervals only, device path: everything that reads the scratch."""
    import torch
    n = int(batch["n_pairs"])
    out = []
    pairs, blocks, ops, off, rc = ctx.refine_host_cigar(batch, flags=0)
    assert rc == 0
    out += [pairs.tobytes(), blocks.tobytes(), ops.tobytes(), off.tobytes()]
    only, rc = ctx.refine_host_intervals(batch)
    assert rc == 0
    out.append(only.tobytes())
    d = api.to_device(batch)
    dp = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
    db = torch.zeros((int(batch["q_len"].astype(np.int64).sum()) + n, 3), dtype=torch.int32, device="cuda:0")
    rc, needed = ctx.refine_device(d, dp, db)
    assert rc == 0
    out += [dp.cpu().numpy().tobytes(), db.cpu().numpy()[:needed].tobytes()]
    return out


@pytest.mark.parametrize("kw", [dict(), dict(band=150), dict(mode=1, band=16)], ids=["lanes", "waves", "affine"])
def test_small_regions_pool_and_repeat_give_the_same_bytes(monkeypatch, synth, kw):
    from blasr_amd import _lib, api
    monkeypatch.setenv("GBA_PART_PAIRS", "300")
    # reads at 25 % error with sparse guides: many short blocks
    batch = synth.compact_blocks(synth.generate(900, read_len=1200, len_jitter=600, p_ins=0.12, p_del=0.08, p_sub=0.05,
                                                drop_frac=0.5, seed=451, decoy_every=5))
    if "mode" in kw:
        kw = dict(kw, mode=_lib.MODE_AFFINE)
    if "band" not in kw or "mode" in kw:
        # one lane per pair however few pairs there are (by default a batch this small runs one wave per pair)
        monkeypatch.setenv("GBA_WAVE_PAIRS", "1")
        monkeypatch.setenv("GBA_WAVE_PAIRS_AFFINE", "1")
    results = {}
    # pool: regions of rows / 64 + 31 blocks and a pool as large as all pairs together; repeat: the same regions and a pool
    # of one pair of the largest size (about seventy of these pairs): it runs out in every part
    for name, env in (("full", {"GBA_BLK_DIV": "1"}), ("default", {}), ("pool", {"GBA_BLK_DIV": "64", "GBA_BLK_POOL_DIV": "1"}),
                      ("repeat", {"GBA_BLK_DIV": "64", "GBA_BLK_POOL_DIV": "1000000"})):
        for k in ("GBA_BLK_DIV", "GBA_BLK_POOL_DIV"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = api.Context(0)
        try:
            ctx.set_scoring(api.make_scoring(**kw))
            results[name] = _run_all(api, _lib, ctx, batch)
            retries = ctx.timing()["scratch_retries"]
            if name == "full":
                wide = ctx_wide_pairs(api, ctx, batch)
                assert (wide < 450) == ("band" not in kw or "mode" in kw)      # lane classes where they were asked for
        finally:
            ctx.close()
        if name == "repeat":
            assert retries >= 1
        else:
            assert retries == 0, name
    pairs = np.frombuffer(results["full"][0], dtype=_lib.PAIR_RESULT_DT)
    ok = pairs["status"] == 0
    assert ok.sum() > 700
    # the small regions are really too small for most pairs (or the test tests nothing): rows / 64 + 31 blocks
    assert (pairs["n_blk"][ok] > pairs["rows"][ok] // 64 + 32).sum() > 300
    for name in ("default", "pool", "repeat"):
        for a, b in zip(results["full"], results[name]):
            assert a == b, name
