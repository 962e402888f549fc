"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo): reads sharded by index, no
data-path collective, results merged in input order.  The per-rank work is done by the oracle
here (this is a test of the host logic; the GPU path is covered by the gpu-marked tests)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from blasr_amd import _lib, shard


def test_shard_indices_partition():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in shard.shard_indices(n, r, world))
            assert seen == list(range(n))


def _oracle_results(oracle, batch):
    """What a context returns for the batch, computed by the oracle: pairs with packed blocks."""
    res, blks = oracle.align_batch(batch, oracle.make_params())
    n = len(res)
    pairs = np.zeros(n, dtype=_lib.PAIR_RESULT_DT)
    for f in ("status", "dp_score", "score", "q_pos", "t_pos", "n_blk", "n_match", "n_mismatch", "n_ins", "n_del", "rows", "cols", "ncells"):
        pairs[f] = res[f]
    nb = np.where(pairs["status"] == 0, pairs["n_blk"], 0).astype(np.int64)
    off = np.zeros(n, np.int64)
    off[1:] = np.cumsum(nb[:-1])
    pairs["blk_off"] = off
    blocks = np.concatenate([b for b, r in zip(blks, res) if r["status"] == 0 and len(b)] or [np.zeros(0, _lib.BLOCK_DT)])
    return pairs, blocks


def test_split_and_merge_keep_reads_together(oracle, synth):
    batch = synth.generate(45, read_len=300, seed=21, decoy_every=5, share_reads=True, ref_len=150_000)
    want_p, want_b = _oracle_results(oracle, batch)
    for world in (1, 2, 4):
        idxs = shard.split_by_read(batch, world)
        assert sorted(np.concatenate(idxs).tolist()) == list(range(45))
        parts = []
        for r, idx in enumerate(idxs):
            assert len(set((batch["read_id"][idx] % world).tolist())) <= 1            # whole reads, all on rank r
            p, b = _oracle_results(oracle, shard.sub_batch(batch, idx))
            parts.append((idx, p, b))
        mp_, mb_ = shard.merge_in_input_order(45, parts)
        assert mp_.tobytes() == want_p.tobytes() and mb_.tobytes() == want_b.tobytes()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_READS, CANDS = 9, 4


def _worker(rank, world, port, name, tmpdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from blasr_amd import synth
    from oracle import oracle
    # this rank's reads of the common set, generated directly (nobody materialises the others')
    n_local = len(shard.shard_indices(N_READS, rank, world)) * CANDS
    local = synth.generate(n_local, read_len=400, seed=9, threads=1, decoy_every=CANDS, share_reads=True,
                           ref_len=150_000, shard=(rank, world))
    pairs, blocks = _oracle_results(oracle, local)
    idx = np.asarray(local["pair_id"])
    m = shard.SharedMerge(name, N_READS * CANDS, rank, create=False, directory=tmpdir)
    m.write(idx, pairs)
    dist.barrier()                                   # timing only: everybody's pairs are in place
    off, total = m.block_offsets()
    if rank == 0:
        m.open_blocks(total, create=True)
    dist.barrier()
    if rank != 0:
        m.open_blocks(total, create=False)
    m.write_blocks(idx, pairs, blocks, off)
    dist.barrier()
    m.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_merge_into_one_array_in_input_order(oracle, synth, tmp_path, world):
    """One process per rank, each with the reads i = rank (mod world) of one common read set and all
    their candidates: the merged result is byte-identical to what one rank computes for the set."""
    name = "t%d" % os.getpid()
    owner = shard.SharedMerge(name, N_READS * CANDS, 0, create=True, directory=str(tmp_path))
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    full = synth.generate(N_READS * CANDS, read_len=400, seed=9, threads=1, decoy_every=CANDS, share_reads=True, ref_len=150_000)
    want_p, want_b = _oracle_results(oracle, full)
    got_p = np.array(owner.pairs)
    owner.open_blocks(len(want_b), create=False)
    got_b = np.array(owner.blocks[:len(want_b)])
    owner.close()
    assert got_p.tobytes() == want_p.tobytes()
    assert got_b.tobytes() == want_b.tobytes()


def test_contiguous_ranges_keep_reads_together_and_results_need_no_gather(oracle, synth):
    """MultiGpuRefiner's partition: contiguous pair ranges cut between reads, balanced by query bases; a range as a
    batch of its own sees only its stretch of the arenas; the ranks' results one after the other are the batch's
    results in input order (Segments.merged: two concatenations and one offset per rank)."""
    batch = synth.generate(60, read_len=300, len_jitter=150, seed=33, decoy_every=5, share_reads=True, ref_len=150_000)
    want_p, want_b = _oracle_results(oracle, batch)
    total = sum(batch[k].nbytes for k in ("q_arena", "t_arena", "blocks"))
    for world in (1, 2, 3, 5, 12, 16):
        ranges = shard.contiguous_ranges(batch, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == 60 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        if world <= 12:
            assert all(hi > lo for lo, hi in ranges)                        # 12 reads: nobody idles
        segs, staged = [], 0
        for lo, hi in ranges:
            if lo > 0 and lo < 60:
                assert batch["read_id"][lo] != batch["read_id"][lo - 1]     # cut between reads
            sub, nbytes = shard.range_batch(batch, lo, hi)
            staged += nbytes
            p, b = _oracle_results(oracle, sub) if hi > lo else (np.zeros(0, _lib.PAIR_RESULT_DT), np.zeros(0, _lib.BLOCK_DT))
            segs.append((lo, hi, p, b, len(b)))
        # shared reads are staged once per rank that has candidates of them; nothing else is staged twice
        assert staged <= total + 1
        s = shard.Segments(60, segs, 0)
        mp_, mb_ = s.merged()
        assert mp_.tobytes() == want_p.tobytes() and mb_.tobytes() == want_b.tobytes()
        r, blk = s.pair(37)
        assert int(r["score"]) == int(want_p[37]["score"])
        assert blk.tobytes() == want_b[int(want_p[37]["blk_off"]):int(want_p[37]["blk_off"]) + int(want_p[37]["n_blk"])].tobytes()
    # balance by query bases: one long read among short ones gets a rank to itself
    q_len = np.array([100] * 10 + [5000] + [100] * 10, np.uint32)
    r = shard.contiguous_ranges({"n_pairs": 21, "q_len": q_len}, 3)
    assert (10, 11) in r
