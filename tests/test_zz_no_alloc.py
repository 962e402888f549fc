"""gba_submit_device_noalloc (the entry point that allocates no device memory) against
gba_submit_device on the same device-resident batches, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(api, _lib, ctx, batch, torch):
    d = api.to_device(batch)
    n = int(batch["n_pairs"])
    cap = int(batch["q_len"].astype(np.int64).sum()) + n
    outs = []
    for no_alloc in (False, True):
        pairs = torch.full((n * _lib.PAIR_RESULT_DT.itemsize,), 0xEE, dtype=torch.uint8, device="cuda:0")
        blocks = torch.full((cap, 3), -1, dtype=torch.int32, device="cuda:0")
        rc, needed = ctx.refine_device(d, pairs, blocks, no_alloc=no_alloc)
        assert rc == 0
        outs.append((pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT), blocks.cpu().numpy()[:needed], needed))
    (p0, b0, n0), (p1, b1, n1) = outs
    assert n0 == n1 and np.array_equal(p0, p1) and np.array_equal(b0, b1)
    assert int((p0["status"] == 0).sum()) >= n - 2
    return p1


def test_no_alloc_entry_point_matches_submit_device(synth):
    import torch
    from blasr_amd import _lib, api
    ctx = api.Context(0)
    try:
        ctx.set_scoring(api.make_scoring())
        big = synth.generate(120, read_len=2400, len_jitter=1200, seed=901, decoy_every=7)
        small = synth.generate(9, read_len=500, len_jitter=200, seed=902)
        _both(api, _lib, ctx, big, torch)
        _both(api, _lib, ctx, small, torch)          # the scratch still holds the first call's bytes
        # wide bands (wave-per-pair kernels) and affine gaps use the same scratch
        ctx.set_scoring(api.make_scoring(band=150))
        _both(api, _lib, ctx, small, torch)
        ctx.set_scoring(api.make_scoring(mode=_lib.MODE_AFFINE, band=16))
        _both(api, _lib, ctx, big, torch)
        # a batch that needs more than the fixed scratch is refused, outputs untouched, and the
        # ordinary entry point still takes it
        ctx.set_scoring(api.make_scoring())
        huge = synth.generate(400, read_len=10_000, seed=903)
        d = api.to_device(huge)
        pairs = torch.full((400 * _lib.PAIR_RESULT_DT.itemsize,), 0xEE, dtype=torch.uint8, device="cuda:0")
        blocks = torch.zeros((int(huge["q_len"].sum()), 3), dtype=torch.int32, device="cuda:0")
        with pytest.raises(_lib.GbaError) as e:
            ctx.refine_device(d, pairs, blocks, no_alloc=True)
        assert e.value.code == _lib.E_TOOBIG and "scratch" in str(e.value)
        assert bool((pairs == 0xEE).all())
        rc, needed = ctx.refine_device(d, pairs, blocks)
        assert rc == 0 and needed > 0
        _both(api, _lib, ctx, small, torch)
    finally:
        ctx.close()


def test_no_alloc_call_leaves_free_device_memory_unchanged(synth):
    """hipMemGetInfo before and after: a call through gba_submit_device_noalloc on a context whose grow-only buffers are
    all still empty does not change the free device memory, while the ordinary entry point on the same context does (so
    the check would see an allocation)."""
    import torch
    from blasr_amd import _lib, api
    warm_ctx = api.Context(0)
    ctx = None
    try:
        def outputs(batch):
            n = int(batch["n_pairs"])
            return (torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0"),
                    torch.zeros((int(batch["q_len"].astype(np.int64).sum()) + n, 3), dtype=torch.int32, device="cuda:0"))
        # first launches load the kernels' code and may allocate on their own: have them behind us
        warm_ctx.set_scoring(api.make_scoring())
        warm = synth.generate(9, read_len=500, len_jitter=200, seed=902)
        warm_ctx.refine_device(api.to_device(warm), *outputs(warm), no_alloc=True)
        ctx = api.Context(0)
        ctx.set_scoring(api.make_scoring())
        big = synth.generate(120, read_len=2400, len_jitter=1200, seed=901)
        d = api.to_device(big)
        pairs, blocks = outputs(big)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        rc, needed = ctx.refine_device(d, pairs, blocks, no_alloc=True)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info(0)[0]
        assert rc == 0 and needed > 0
        assert free1 == free0, (free0, free1)
        want = pairs.cpu().numpy().copy()
        rc, _ = ctx.refine_device(d, pairs, blocks)
        torch.cuda.synchronize()
        assert rc == 0 and torch.cuda.mem_get_info(0)[0] < free1          # the ordinary path grows the context's buffers
        assert np.array_equal(pairs.cpu().numpy(), want)
    finally:
        warm_ctx.close()
        if ctx is not None:
            ctx.close()
