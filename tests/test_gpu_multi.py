"""The multi-GPU scheduler on one GPU: two (three) contexts on cuda:0 stand in for as many GPUs."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_multi_gpu_refiner_equals_one_context(oracle, synth, world):
    """One batch, a contiguous range of reads (all their candidates with them) per context, every context
    running at the same time: the merged result is byte-identical to one context's, whatever the
    number of contexts (the reference's multi-thread output needs a sort to compare,
    /root/reference/ctest/fofn.t:17); every context stages its own share of the arenas only."""
    from blasr_amd import api, shard
    ref = synth.reference(800_000, n_frac=0.01)
    batch = synth.generate(330, read_len=1800, len_jitter=600, seed=77, decoy_every=6, share_reads=True, ref=ref)
    one = api.Context(0)
    one.set_scoring(api.make_scoring())
    want_p, want_b, rc = one.refine_host(batch)
    assert rc == 0
    util.compare_batch(oracle, batch, oracle.make_params(), want_p, want_b)
    m = shard.MultiGpuRefiner([0] * world)
    try:
        m.set_scoring(api.make_scoring())
        got_p, got_b, rc = m.refine_host(batch)
        assert rc == 0
        assert got_p.tobytes() == want_p.tobytes() and got_b.tobytes() == want_b.tobytes()
        total = sum(batch[k].nbytes for k in ("q_arena", "t_arena", "blocks"))
        assert sum(m.bytes_staged) <= total and max(m.bytes_staged) < 0.6 * total * 2 / world        # a share each, not the whole batch
        # a second batch reuses the pinned result buffers, sized from what the first one needed; the segments in order are
        # the batch in input order without any copy
        segs = m.refine_host_segments(batch)
        assert segs.rc == 0 and [s[0] for s in segs] == [r[0] for r in shard.contiguous_ranges(batch, world)]
        for p in (0, 137, 329):
            r, blk = segs.pair(p)
            assert r.tobytes() [:24] == want_p[p].tobytes()[:24] and int(r["score"]) == int(want_p[p]["score"])
            assert blk.tobytes() == want_b[int(want_p[p]["blk_off"]):int(want_p[p]["blk_off"]) + int(want_p[p]["n_blk"])].tobytes()
        # the same through windows of a reference every context holds
        m.set_reference(ref)
        got_p, got_b, rc = m.refine_host(synth.as_windows(batch) | {"read_id": batch["read_id"]})
        assert rc == 0 and got_p.tobytes() == want_p.tobytes() and got_b.tobytes() == want_b.tobytes()
        # and from the sequences alone: every context makes the guides of its own reads
        seqs = {k: v for k, v in batch.items() if k not in ("blocks", "blk_off", "n_blk")}
        sp = api.make_sdp_params()
        one.set_guide_producer(sp)
        want_p, want_b, rc = one.refine_host(seqs)
        assert rc == 0 and (want_p["status"] == 0).sum() >= 55
        m.set_guide_producer(sp)
        got_p, got_b, rc = m.refine_host(seqs)
        assert rc == 0 and got_p.tobytes() == want_p.tobytes() and got_b.tobytes() == want_b.tobytes()
    finally:
        m.close()
        one.close()


def test_two_contexts_per_worker_overlap_batches(synth):
    """depth 2: a second batch is submitted while the first one runs (its inputs travel under the first one's kernels); both
    come back complete and in order, equal to what one context returns; a block arena guessed too small is run again at size."""
    from blasr_amd import api, shard
    a = synth.generate(200, read_len=1500, len_jitter=500, seed=5)
    b = synth.generate(260, read_len=900, len_jitter=300, seed=6)
    one = api.Context(0)
    one.set_scoring(api.make_scoring())
    want = [one.refine_host(x) for x in (a, b)]
    m = shard.MultiGpuRefiner([0, 0], depth=2, stagger=0.0)
    try:
        m.set_scoring(api.make_scoring())
        m._last_needed = [8, 8]                      # as if the last batch had needed next to nothing: forces the retry
        m.submit(a); m.submit(b)
        for w in want:
            s = m.wait()
            gp, gb = s.merged()
            assert s.rc == 0 and gp.tobytes() == w[0].tobytes() and gb.tobytes() == w[1].tobytes()
        # interval output: no blocks anywhere
        s = m.refine_host_segments(a, want_blocks=False)
        gp, gb = s.merged()
        assert s.rc == 0 and len(gb) == 0 and (gp["score"] == want[0][0]["score"]).all() and (gp["q_end"] == want[0][0]["q_end"]).all()
    finally:
        m.close()
        one.close()


def test_device_pipeline_equals_one_context(synth):
    """Device-resident batches streamed through two contexts on one GPU, a worker thread each (api.DevicePipeline): every
    batch comes back with the bytes one context gives, in submission order, whatever overlaps with what."""
    import torch
    from blasr_amd import _lib, api
    batches = [synth.compact_blocks(synth.generate(n, read_len=rl, len_jitter=rl // 2, seed=900 + i))
               for i, (n, rl) in enumerate(((500, 1500), (64, 4000), (900, 700), (1, 300), (300, 2500)))]
    dev = [api.to_device(b) for b in batches]
    outs = [(torch.zeros(int(b["n_pairs"]) * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0"),
             torch.zeros((int(b["q_len"].astype(np.int64).sum()) + int(b["n_pairs"]), 3), dtype=torch.int32, device="cuda:0")) for b in batches]
    one = api.Context(0)
    one.set_scoring(api.make_scoring())
    want = []
    for d, o in zip(dev, outs):
        rc, needed = one.refine_device(d, *o)
        assert rc == 0
        want.append((o[0].cpu().numpy().tobytes(), o[1][:needed].cpu().numpy().tobytes(), needed))
        o[0].zero_(); o[1].zero_()
    one.close()
    pipe = api.DevicePipeline(0, depth=2)
    try:
        pipe.set_scoring(api.make_scoring())
        with pytest.raises(api.GbaError):
            pipe.wait()                                   # nothing in flight
        got = []
        for k in range(len(dev)):
            if k >= 2:
                got.append(pipe.wait())
            pipe.submit(dev[k], *outs[k])
        with pytest.raises(api.GbaError):
            pipe.submit(dev[0], *outs[0])                 # two in flight already
        got.append(pipe.wait()); got.append(pipe.wait())
        for k, (rc, needed) in enumerate(got):
            assert rc == 0 and needed == want[k][2]
            assert outs[k][0].cpu().numpy().tobytes() == want[k][0]
            assert outs[k][1][:needed].cpu().numpy().tobytes() == want[k][1]
    finally:
        pipe.close()
