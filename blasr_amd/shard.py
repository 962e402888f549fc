"""Multi-GPU scheduling of the refine path.

Reads are independent, so the reads with index = r (mod world) go to GPU r -- what BLASR's own
-start/-stride does across processes (/root/reference/iblasr/RegisterBlasrOptions.h:93-94,
Blasr.cpp:1270) -- all candidates of a read stay on one GPU (the per-read steps that follow the
refine, iblasr/BlasrAlignImpl.hpp:381-382 and the sort at iblasr/BlasrUtilsImpl.hpp:920-922, then
need nothing from another GPU), and the results are merged on the host in input order, so the
output does not depend on the number of GPUs.  There is no collective on the data path.

Two forms:
  MultiGpuRefiner   one process, one pipeline of contexts per GPU: the replacement of the reference's
                    pool of -nproc worker threads (Blasr.cpp:1457-1488).  One caller hands over one
                    batch; GPU g gets a CONTIGUOUS range of its reads (cut at read boundaries, balanced by
                    query bases), so it stages and holds only its own share of the arenas, and the results
                    in input order are the GPUs' result arrays one after the other: nothing is gathered.
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


def contiguous_ranges(batch, world: int) -> List[tuple]:
    """(lo, hi) pair ranges, one per rank, that tile the batch in order: cut only between reads
    (batch["read_id"]; every pair its own read without it) and balanced by query bases, the
    predictor of a pair's DP cells (rows x band)."""
    n = int(batch["n_pairs"])
    if n == 0:
        return [(0, 0)] * world
    rid = np.asarray(batch["read_id"]) if batch.get("read_id") is not None else np.arange(n)
    cost = np.cumsum(np.asarray(batch["q_len"], np.int64))
    starts = np.flatnonzero(np.concatenate(([True], rid[1:] != rid[:-1])))        # first pair of every read
    before = np.concatenate(([0], cost))[starts]                                   # cost in front of each read
    cuts, k_prev, R = [0], 0, len(starts)
    for r in range(1, world):
        want = cost[-1] * r / world
        k = int(np.searchsorted(before, want, side="left"))          # the read boundary nearest to the rank's share
        if 0 < k < R and want - before[k - 1] < before[k] - want:
            k -= 1
        if R >= world:                                  # every rank gets a read when there are enough of them
            k = min(max(k, k_prev + 1), R - (world - r))
        k = max(k, k_prev)
        cuts.append(int(starts[k]) if k < R else n)
        k_prev = k
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


_ARENAS = (("q_arena", "q_off", "q_len", 1), ("t_arena", "t_off", "t_len", 1), ("blocks", "blk_off", "n_blk", 1))


def range_batch(batch, lo: int, hi: int):
    """Pairs [lo, hi) of a batch as a batch of their own whose arenas are the stretches those pairs point
    into (views, nothing is copied but the three offset arrays, which are rebased): what one GPU of
    several stages, holds and is told the extent of.  Returns (sub-batch, bytes of the three stretches)."""
    out = {"n_pairs": hi - lo}
    for k, v in batch.items():
        if k in PER_PAIR and v is not None:
            out[k] = np.asarray(v)[lo:hi]
        elif k != "n_pairs":
            out[k] = v
    staged = 0
    packed = bool(batch.get("q_packed"))
    for arena, off, length, _ in _ARENAS:
        a = batch.get(arena)
        if a is None or hi == lo:
            continue
        o = np.asarray(batch[off][lo:hi], np.int64); l = np.asarray(batch[length][lo:hi], np.int64)
        live = l > 0
        if not live.any():
            continue
        first, last = int(o[live].min()), int((o[live] + l[live]).max())
        if arena == "q_arena" and packed:
            first -= first % 16                         # packed reads move 16 bases (4 bytes) at a time
            out[arena] = np.asarray(a)[first // 4:(last + 15) // 16 * 4]
            out["q_bases"] = last - first
        else:
            out[arena] = np.asarray(a)[first:last]
        out[off] = (np.asarray(batch[off][lo:hi], np.uint64) - np.uint64(first)).astype(np.uint64)
        if arena == "q_arena" and batch.get("q_profile") is not None:
            out["q_profile"] = np.asarray(batch["q_profile"])[first * 11:last * 11]
        staged += out[arena].nbytes
    return out, staged


class Segments:
    """The results of one batch as the GPUs left them: segment g = (lo, hi, pairs, blocks, blocks_needed)
    for the pairs [lo, hi) of the batch; pairs[k]["blk_off"] indexes the segment's own block array.
    Segments tile the batch in order, so reading them one after the other is reading the batch in
    input order; merged() copies them into the two arrays a single context would have returned."""

    def __init__(self, n, segs, rc):
        self.n, self.segs, self.rc = n, segs, rc

    def __iter__(self):
        return iter(self.segs)

    def pair(self, p):
        """(pair result, its blocks) of input pair p."""
        for lo, hi, pairs, blocks, _ in self.segs:
            if lo <= p < hi:
                r = pairs[p - lo]
                return r, (blocks[int(r["blk_off"]):int(r["blk_off"]) + int(r["n_blk"])] if blocks is not None and int(r["status"]) == 0 else None)
        raise IndexError(p)

    def merged(self):
        """(pairs, blocks) of the whole batch, blocks packed in input order, every blk_off rewritten:
        two sequential copies and one prefix per GPU added to its pairs' offsets."""
        pairs = np.concatenate([s[2] for s in self.segs]) if self.segs else np.zeros(0, _lib.PAIR_RESULT_DT)
        want = all(s[3] is not None for s in self.segs)
        blocks = np.concatenate([s[3][:s[4]] for s in self.segs]) if want and self.segs else np.zeros(0, _lib.BLOCK_DT)
        base = 0
        for lo, hi, _, _, needed in self.segs:
            pairs["blk_off"][lo:hi] += np.uint64(base)
            base += needed
        return pairs, blocks


class MultiGpuRefiner:
    """One caller, several GPUs, one batch after the other, results in input order.

    devices: the GPUs (a GPU may be named more than once: each entry is a worker of its own).  depth: contexts
    per entry, used in turn (api.HostPipeline): with 2, batch k + 1 can be submitted while batch k runs, its
    inputs travel under batch k's kernels.  Result buffers are page-locked memory of the context that fills
    them, kept and reused from batch to batch; the block arena of a GPU is sized from what its last batch
    needed (a first batch: a block per four query bases) and a batch that outgrows it is run again with the
    size it reported."""

    def __init__(self, devices: Sequence[int], depth: int = 1, stagger: float = 0.4):
        from .api import HostPipeline
        self.pipes = [HostPipeline(d, depth=depth, stagger=stagger) for d in devices]
        self.depth = depth
        self._bufs = [[{} for _ in range(depth)] for _ in devices]       # per worker and slot: pinned result buffers
        self._flight = []                                                # tickets, oldest first
        self._sub = 0
        self._last_needed = [0] * len(devices)
        self.bytes_staged = [0] * len(devices)                           # arena bytes of the last batch, per worker

    @property
    def ctxs(self):
        return [c for p in self.pipes for c in p.ctxs]

    def close(self):
        for p in self.pipes:
            p.close()

    def set_scoring(self, scoring):
        for p in self.pipes:
            p.set_scoring(scoring)

    def set_reference(self, ref):
        for p in self.pipes:
            p.set_reference(ref)

    def set_guide_producer(self, params):
        """Batches without "blocks" get their guides on the GPU that refines them (Context.set_guide_producer)."""
        for p in self.pipes:
            p.set_guide_producer(params)

    def _buffer(self, g, slot, name, count, dtype):
        have = self._bufs[g][slot].get(name)
        if have is None or len(have) < count:
            have = self.pipes[g].ctxs[slot].host_array(max(count, 1) + count // 8, dtype)
            self._bufs[g][slot][name] = have
        return have[:count]

    def _submit_range(self, g, slot, sub, want_blocks, cap):
        n = int(sub["n_pairs"])
        pairs = self._buffer(g, slot, "pairs", n, _lib.PAIR_RESULT_DT)
        blocks = self._buffer(g, slot, "blocks", cap, _lib.BLOCK_DT) if want_blocks else None
        self.pipes[g].submit(sub, pairs, blocks)
        return pairs, blocks

    def submit(self, batch, want_blocks: bool = True):
        """Starts the batch on every GPU (gba_submit does not block) and returns; wait() collects it."""
        if len(self._flight) >= self.depth:
            raise _lib.GbaError(_lib.E_STATE)
        slot = self._sub % self.depth
        parts = []
        for g, (lo, hi) in enumerate(contiguous_ranges(batch, len(self.pipes))):
            sub, staged = range_batch(batch, lo, hi)
            self.bytes_staged[g] = staged
            qsum = int(np.asarray(sub["q_len"], np.int64).sum()) if hi > lo else 0
            cap = self._last_needed[g] + self._last_needed[g] // 8 + 64 if self._last_needed[g] else qsum // 4 + (hi - lo) + 64
            pairs, blocks = self._submit_range(g, slot, sub, want_blocks, cap) if hi > lo else (np.zeros(0, _lib.PAIR_RESULT_DT), None)
            parts.append([lo, hi, sub, pairs, blocks, slot])
        self._flight.append((int(batch["n_pairs"]), parts, want_blocks))
        self._sub += 1

    def wait(self) -> Segments:
        """Blocks until the oldest submitted batch is complete on every GPU."""
        if not self._flight:
            raise _lib.GbaError(_lib.E_STATE)
        n, parts, want_blocks = self._flight.pop(0)
        segs, rc_all = [], 0
        for g, (lo, hi, sub, pairs, blocks, slot) in enumerate(parts):
            if hi == lo:
                segs.append((lo, hi, pairs, blocks if want_blocks else None, 0))
                continue
            rc, needed = self.pipes[g].wait()
            if rc == _lib.E_NOSPACE and want_blocks:         # the arena guessed from the last batch was too small: once more, at size
                blocks = self._buffer(g, slot, "blocks", needed + 64, _lib.BLOCK_DT)
                rc, needed = self.pipes[g].rerun_last(sub, pairs, blocks)
            self._last_needed[g] = needed
            rc_all = rc_all or rc
            segs.append((lo, hi, pairs, blocks, needed))
        return Segments(n, segs, rc_all)

    def refine_host_segments(self, batch, want_blocks: bool = True) -> Segments:
        self.submit(batch, want_blocks)
        return self.wait()

    def refine_host(self, batch):
        """Returns (pairs, blocks, rc) like Context.refine_host, for the whole batch (Segments.merged())."""
        s = self.refine_host_segments(batch)
        mp, mb = s.merged()
        return mp, mb, s.rc


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
