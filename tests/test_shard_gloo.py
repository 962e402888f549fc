"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo): reads sharded by index, no
data-path collective, results merged in input order.  The per-rank work is done by the oracle
here (this is a test of the host logic; the GPU path is covered by the gpu-marked tests)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from blasr_amd import shard


def test_shard_indices_partition():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in shard.shard_indices(n, r, world))
            assert seen == list(range(n))
    per_rank = [[("r", i) for i in shard.shard_indices(10, r, 3)] for r in range(3)]
    assert shard.merge_in_input_order(10, per_rank) == [("r", i) for i in range(10)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_reads, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from blasr_amd import synth
    from oracle import oracle
    batch = synth.generate(n_reads, read_len=400, seed=9, threads=1)
    P = oracle.make_params()
    local = []
    for i in shard.shard_indices(n_reads, rank, world):
        qq, tt, blk = synth.pair(batch, i)
        r, b = oracle.align(qq, tt, blk, P)
        local.append((i, r["score"], r["n_blk"]))
    merged = shard.gather_results(local, n_reads)
    if rank == 0:
        q.put(merged)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_merge_in_input_order():
    from blasr_amd import synth
    from oracle import oracle
    synth.build(); oracle.build()
    n_reads, world, port = 13, 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [m[0] for m in merged] == list(range(n_reads))
    batch = synth.generate(n_reads, read_len=400, seed=9, threads=1)
    P = oracle.make_params()
    for i in range(n_reads):
        qq, tt, blk = synth.pair(batch, i)
        r, _ = oracle.align(qq, tt, blk, P)
        assert merged[i] == (i, r["score"], r["n_blk"])
