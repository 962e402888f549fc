"""Multi-GPU sharding of the refine path: reads are independent, so rank r owns the reads with
index = r (mod world) -- what BLASR's own -start/-stride does across processes
(/root/reference/iblasr/RegisterBlasrOptions.h:93-94, Blasr.cpp:1270) -- all candidates of a
read stay on one rank, and results are merged on the host in input order.  No data-path
collective exists; the only communication is the final gather of per-read results."""
from typing import Any, List, Sequence


def shard_indices(n_reads: int, rank: int, world: int) -> range:
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    return range(rank, n_reads, world)


def merge_in_input_order(n_reads: int, per_rank: Sequence[Sequence[Any]]) -> List[Any]:
    """per_rank[r][k] is the result of read shard_indices(n, r, world)[k]."""
    world = len(per_rank)
    out: List[Any] = [None] * n_reads
    for r, results in enumerate(per_rank):
        idx = shard_indices(n_reads, r, world)
        if len(results) != len(idx):
            raise ValueError("rank %d returned %d results for %d reads" % (r, len(results), len(idx)))
        for k, i in enumerate(idx):
            out[i] = results[k]
    return out


def gather_results(local_results: Sequence[Any], n_reads: int):
    """Gather every rank's per-read results on rank 0 (torch.distributed, any backend) and merge
    them in input order.  Returns the merged list on rank 0, None elsewhere."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(list(local_results), gathered, dst=0)
    return merge_in_input_order(n_reads, gathered) if rank == 0 else None
