"""Multi-GPU scheduling of the refine path.

Reads are independent, so the reads with index = r (mod world) go to GPU r -- what BLASR's own
-start/-stride does across processes (/root/reference/iblasr/RegisterBlasrOptions.h:93-94,
Blasr.cpp:1270) -- all candidates of a read stay on one GPU (the per-read steps that follow the
refine, iblasr/BlasrAlignImpl.hpp:381-382 and the sort at iblasr/BlasrUtilsImpl.hpp:920-922, then
need nothing from another GPU), and the results are merged on the host in input order, so the
output does not depend on the number of GPUs.  There is no collective on the data path.

Two forms:
  MultiGpuRefiner   one process, one context per GPU: the replacement of the reference's pool of
                    -nproc worker threads (Blasr.cpp:1457-1488).  One caller hands over one batch;
                    every GPU gets its reads through the asynchronous gba_submit, all run at the
                    same time, the results come back as one array in input order.
  SharedMerge       one process per GPU (torch.distributed launch): every rank writes the results
                    of its reads into one node-wide, preallocated array at the reads' own places.
"""
import os
from typing import List, Sequence

import numpy as np

from . import _lib

PER_PAIR = ("q_off", "q_len", "t_off", "t_len", "blk_off", "n_blk", "t_rc", "win_start", "win_rc", "read_id", "pair_id")


def shard_indices(n_reads: int, rank: int, world: int) -> range:
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    return range(rank, n_reads, world)


def split_by_read(batch, world: int) -> List[np.ndarray]:
    """Pair indices per rank: the pairs of read i (batch["read_id"], else every pair is its own
    read) go to rank i % world, in input order."""
    n = int(batch["n_pairs"])
    rid = np.asarray(batch["read_id"]) if batch.get("read_id") is not None else np.arange(n)
    owner = rid % world
    return [np.flatnonzero(owner == r) for r in range(world)]


def sub_batch(batch, idx: np.ndarray):
    """The pairs `idx` of a batch as a batch of their own.  Arenas are shared, not copied: only the
    per-pair arrays are gathered."""
    out = {"n_pairs": int(len(idx))}
    for k, v in batch.items():
        if k == "n_pairs":
            continue
        out[k] = np.ascontiguousarray(np.asarray(v)[idx]) if k in PER_PAIR and v is not None else v
    return out


def merge_in_input_order(n: int, parts: Sequence):
    """parts: per rank (idx, pairs, blocks) with blk_off indexing that rank's own block arena.
    Returns (pairs, blocks) of the whole batch in input order, blocks packed in input order, every
    blk_off rewritten to index the merged arena: exactly what one context returns for the batch."""
    pairs = np.zeros(n, dtype=_lib.PAIR_RESULT_DT)
    for idx, p, _ in parts:
        pairs[idx] = p
    nb = np.where(pairs["status"] == _lib.PAIR_OK, pairs["n_blk"], 0).astype(np.int64)
    new_off = np.zeros(n, np.int64)
    new_off[1:] = np.cumsum(nb[:-1])
    blocks = np.zeros(int(nb.sum()), dtype=_lib.BLOCK_DT)
    for idx, p, b in parts:
        cnt = nb[idx]
        tot = int(cnt.sum())
        if tot == 0:
            continue
        src0 = p["blk_off"].astype(np.int64)
        src_start = np.zeros(len(idx), np.int64)
        src_start[1:] = np.cumsum(cnt[:-1])
        within = np.arange(tot, dtype=np.int64) - np.repeat(src_start, cnt)
        blocks[np.repeat(new_off[idx], cnt) + within] = b[np.repeat(src0, cnt) + within]
    pairs["blk_off"] = new_off.astype(np.uint64)
    return pairs, blocks


class MultiGpuRefiner:
    """One caller, several GPUs, one batch, results in input order."""

    def __init__(self, devices: Sequence[int]):
        from .api import Context
        self.ctxs = [Context(d) for d in devices]

    def close(self):
        for c in self.ctxs:
            c.close()

    def set_scoring(self, scoring):
        for c in self.ctxs:
            c.set_scoring(scoring)

    def set_reference(self, ref):
        for c in self.ctxs:
            c.set_reference(ref)

    def set_guide_producer(self, params):
        """Batches without "blocks" get their guides on the GPU that refines them (Context.set_guide_producer)."""
        for c in self.ctxs:
            c.set_guide_producer(params)

    def refine_host(self, batch):
        """Returns (pairs, blocks, rc) like Context.refine_host, for the whole batch."""
        n = int(batch["n_pairs"])
        idxs = split_by_read(batch, len(self.ctxs))
        flight = []
        for c, idx in zip(self.ctxs, idxs):           # every GPU starts at once: gba_submit does not block
            sb = sub_batch(batch, idx)
            cap = int(np.asarray(sb["q_len"]).astype(np.int64).sum()) + len(idx)
            pairs = np.zeros(len(idx), dtype=_lib.PAIR_RESULT_DT)
            blocks = np.zeros(cap, dtype=_lib.BLOCK_DT)
            c.submit_host(sb, pairs, blocks)
            flight.append((idx, pairs, blocks))
        rc = 0
        for c in self.ctxs:
            r, _ = c.wait_host()
            rc = rc or r
        mp, mb = merge_in_input_order(n, flight)
        return mp, mb, rc


class SharedMerge:
    """Node-wide result array for one process per GPU: rank r writes the results of its reads at
    their places in input order; nobody sends anything to anybody.  The array is a file in shared
    memory that every rank maps; the rank that creates it (rank 0) removes it in close()."""

    def __init__(self, name: str, n_total: int, rank: int, create: bool, directory: str = "/dev/shm"):
        self.path = os.path.join(directory, "gba_merge_%s.pairs" % name)
        self.owner = create
        self.rank = rank
        mode = "w+" if create else "r+"
        self.pairs = np.memmap(self.path, dtype=_lib.PAIR_RESULT_DT, mode=mode, shape=(max(n_total, 1),))
        self.n = n_total

    def write(self, idx: np.ndarray, pairs: np.ndarray):
        """Phase 1: this rank's pair results (blk_off still indexes the rank's own block arena)."""
        self.pairs[idx] = pairs
        self.pairs.flush()

    def block_offsets(self):
        """After every rank has written its pairs (the caller's barrier): where each pair's blocks
        go in the merged arena, and the arena's size.  Every rank computes the same numbers."""
        nb = np.where(self.pairs["status"] == _lib.PAIR_OK, self.pairs["n_blk"], 0).astype(np.int64)
        off = np.zeros(self.n, np.int64)
        off[1:] = np.cumsum(nb[:-1])
        return off, int(nb.sum())

    def open_blocks(self, total: int, create: bool):
        """Phase 2 (rank 0 with create=True first, the others after another barrier)."""
        self.bpath = self.path[:-6] + ".blocks"
        self.blocks = np.memmap(self.bpath, dtype=_lib.BLOCK_DT, mode="w+" if create else "r+", shape=(max(total, 1),))

    def write_blocks(self, idx: np.ndarray, pairs: np.ndarray, blocks: np.ndarray, off: np.ndarray):
        cnt = np.where(pairs["status"] == _lib.PAIR_OK, pairs["n_blk"], 0).astype(np.int64)
        tot = int(cnt.sum())
        if tot:
            start = np.zeros(len(idx), np.int64)
            start[1:] = np.cumsum(cnt[:-1])
            within = np.arange(tot, dtype=np.int64) - np.repeat(start, cnt)
            self.blocks[np.repeat(off[idx], cnt) + within] = blocks[np.repeat(pairs["blk_off"].astype(np.int64), cnt) + within]
        self.blocks.flush()
        self.pairs["blk_off"][idx] = off[idx].astype(np.uint64)
        self.pairs.flush()

    def close(self):
        del self.pairs
        if hasattr(self, "blocks"):
            del self.blocks
            if self.owner and os.path.exists(self.bpath):
                os.remove(self.bpath)
        if self.owner and os.path.exists(self.path):
            os.remove(self.path)
