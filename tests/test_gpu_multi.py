"""The multi-GPU scheduler on one GPU: two (three) contexts on cuda:0 stand in for as many GPUs."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_multi_gpu_refiner_equals_one_context(oracle, synth, world):
    """One batch, reads dealt round-robin over the contexts with all their candidates, every context
    running at the same time: the merged result is byte-identical to one context's, whatever the
    number of contexts (the reference's multi-thread output needs a sort to compare,
    /root/reference/ctest/fofn.t:17)."""
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
