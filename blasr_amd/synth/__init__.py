"""Synthetic (read, reference-window, guide) pair batches.

Thin ctypes wrapper over synth.c (built in-tree as libgbasynth.so).  A batch is a
dict of numpy arrays in the struct-of-arrays layout the C-ABI takes
(include/gba.h, gba_batch): byte arenas plus per-pair offsets and lengths.
"""
import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgbasynth.so")

BLOCK_DT = np.dtype([("q_pos", "<u4"), ("t_pos", "<u4"), ("length", "<u4")])

ECOLI_LEN = 4_641_652      # E. coli K-12 MG1655
CHR20_LEN = 64_444_167     # GRCh38 chr20
REF_SEED = 20250218


class _Cfg(C.Structure):
    _fields_ = [("read_len", C.c_uint32), ("p_ins", C.c_double), ("p_del", C.c_double),
                ("p_sub", C.c_double), ("min_block", C.c_uint32), ("drop_frac", C.c_double),
                ("seed", C.c_uint64), ("q_stride", C.c_uint32), ("t_stride", C.c_uint32),
                ("blk_stride", C.c_uint32), ("decoy_every", C.c_uint32), ("len_jitter", C.c_uint32),
                ("shard_rank", C.c_uint32), ("shard_world", C.c_uint32), ("share_reads", C.c_uint32)]


def build(force=False):
    src = os.path.join(_HERE, "synth.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", _SO, src])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.gbs_generate.restype = C.c_uint32
        _lib.gbs_reference.restype = None
    return _lib


_ref_cache = {}


def reference(length=ECOLI_LEN, seed=REF_SEED, n_frac=0.0, n_run=1000):
    key = (length, seed, n_frac, n_run)
    if key not in _ref_cache:
        ref = np.empty(length, dtype=np.uint8)
        _load().gbs_reference(ref.ctypes.data_as(C.c_void_p), C.c_uint64(length), C.c_uint64(seed),
                              C.c_double(n_frac), C.c_uint32(n_run))
        _ref_cache[key] = ref
    return _ref_cache[key]


# Workload presets.  Error split and lengths follow the measurement plan for the
# five benchmark configurations; `band`/`mode` are what the caller should align with.
PRESETS = {
    # plumbing: 1-3 kb subread-like pairs
    "c1_ecoli_plumbing": dict(read_len=3000, len_jitter=2000, p_ins=0.09, p_del=0.04, p_sub=0.02,
                              ref_len=ECOLI_LEN, seed=1000, band=10, mode="linear"),
    # 10 kb, 15 % error, indel-heavy, linear gaps, band 10
    "c2_10kb_linear": dict(read_len=10000, p_ins=0.09, p_del=0.04, p_sub=0.02,
                           ref_len=ECOLI_LEN, seed=2000, band=10, mode="linear"),
    # 20 kb, affine gaps, band 16
    "c3_20kb_affine": dict(read_len=20000, p_ins=0.09, p_del=0.04, p_sub=0.02,
                           ref_len=ECOLI_LEN, seed=3000, band=16, mode="affine"),
    # chr20-sized reference with N runs, 10 candidates per read (1 true + 9 decoys)
    "c4_chr20_bestn10": dict(read_len=10000, p_ins=0.09, p_del=0.04, p_sub=0.02,
                             ref_len=CHR20_LEN, n_frac=0.01, seed=4000, band=10, mode="linear",
                             decoy_every=10),
    # 50 kb, 20 % error, wide band
    "c5_50kb_wide": dict(read_len=50000, p_ins=0.11, p_del=0.06, p_sub=0.03,
                         ref_len=ECOLI_LEN, seed=5000, band=100, mode="linear"),
}


def generate(n_pairs, read_len=10000, p_ins=0.09, p_del=0.04, p_sub=0.02, min_block=11,
             drop_frac=0.10, seed=2000, ref_len=ECOLI_LEN, n_frac=0.0, decoy_every=0,
             len_jitter=0, threads=8, ref=None, share_reads=False, shard=(0, 1), **_ignored):
    """Generate `n_pairs` pairs.  Returns a dict with q_arena, q_off, q_len, t_arena,
    t_off, t_len, blocks, blk_off, n_blk, n_pairs, plus win_start / win_rc (each target as a
    window of the reference: start on the forward strand, strand flag) and read_id (pairs with
    the same id are candidates of one read).  With decoy_every = k, pairs p with p % k != 0 are
    decoys: random loci with sparse guides; with share_reads they are further candidates of the
    true pair's read (they view its query bytes), otherwise unrelated random reads.
    shard = (rank, world): the pairs generated are those of the reads with index = rank (mod world)
    of the common set that shard=(0, 1) generates, in order (a read = decoy_every consecutive pairs,
    or one pair); read_id then holds the reads' indices in the common set."""
    lib = _load()
    if ref is None:
        ref = reference(ref_len, REF_SEED, n_frac)
    q_stride = int(read_len)
    t_stride = int(read_len * 1.35) + 256
    blk_stride = read_len // min_block + 2
    cfg = _Cfg(read_len, p_ins, p_del, p_sub, min_block, drop_frac, seed, q_stride, t_stride,
               blk_stride, decoy_every, len_jitter, int(shard[0]), int(shard[1]), 1 if share_reads else 0)
    b = {
        "n_pairs": n_pairs,
        "q_arena": np.zeros(n_pairs * q_stride, dtype=np.uint8),
        "t_arena": np.zeros(n_pairs * t_stride, dtype=np.uint8),
        "blocks": np.zeros(n_pairs * blk_stride, dtype=BLOCK_DT),
        "q_off": np.zeros(n_pairs, dtype=np.uint64), "q_len": np.zeros(n_pairs, dtype=np.uint32),
        "t_off": np.zeros(n_pairs, dtype=np.uint64), "t_len": np.zeros(n_pairs, dtype=np.uint32),
        "blk_off": np.zeros(n_pairs, dtype=np.uint64), "n_blk": np.zeros(n_pairs, dtype=np.uint32),
        "win_start": np.zeros(n_pairs, dtype=np.uint64), "win_rc": np.zeros(n_pairs, dtype=np.uint8),
    }
    idx = np.arange(n_pairs, dtype=np.int64)
    k = decoy_every if (share_reads and decoy_every) else 1
    kk = decoy_every if decoy_every else 1
    # reads of this shard in the common set: groups of kk pairs are dealt round-robin over the ranks
    common = ((idx // kk) * int(shard[1]) + int(shard[0])) * kk + idx % kk
    b["pair_id"] = common                              # index of each pair in the common set
    b["read_id"] = common // k
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def run(first, count):
        return lib.gbs_generate(p(ref), C.c_uint64(len(ref)), C.byref(cfg), C.c_uint32(first),
                                C.c_uint32(count), p(b["q_arena"]), p(b["q_off"]), p(b["q_len"]),
                                p(b["t_arena"]), p(b["t_off"]), p(b["t_len"]),
                                p(b["blocks"]), p(b["blk_off"]), p(b["n_blk"]), p(b["win_start"]), p(b["win_rc"]))

    threads = max(1, min(threads, n_pairs))
    chunk = (n_pairs + threads - 1) // threads
    if decoy_every:
        chunk = (chunk + decoy_every - 1) // decoy_every * decoy_every     # a read's candidates in one job
    jobs = [(i, min(chunk, n_pairs - i)) for i in range(0, n_pairs, chunk)]
    if len(jobs) == 1:
        bad = [run(*jobs[0])]
    else:
        with ThreadPoolExecutor(len(jobs)) as ex:
            bad = list(ex.map(lambda j: run(*j), jobs))
    if any(bad):
        raise RuntimeError("synthetic pair %d overflowed its block stride" % (max(bad) - 1))
    return b


def preset(name, n_pairs, **over):
    cfg = dict(PRESETS[name])
    cfg.update(over)
    return generate(n_pairs, **cfg), cfg


def compact_blocks(batch):
    """The same batch with its guide blocks packed end to end (the generator leaves every pair a
    fixed stride of the block arena, most of it unused)."""
    nb = np.asarray(batch["n_blk"], dtype=np.int64)
    new_off = np.zeros(len(nb), np.int64)
    new_off[1:] = np.cumsum(nb[:-1])
    total = int(nb.sum())
    src = np.repeat(np.asarray(batch["blk_off"], dtype=np.int64) - new_off, nb) + np.arange(total, dtype=np.int64)
    out = dict(batch)
    out["blocks"] = np.ascontiguousarray(batch["blocks"][src]) if total else np.zeros(1, BLOCK_DT)
    out["blk_off"] = new_off.astype(np.uint64)
    return out


def as_windows(batch):
    """The same batch with its targets given as windows of the reference (for a context that holds
    the reference: no t_arena; t_off = window start on the forward strand, t_rc = strand)."""
    w = {k: batch[k] for k in ("n_pairs", "q_arena", "q_off", "q_len", "t_len", "blocks", "blk_off", "n_blk")}
    w["t_off"] = batch["win_start"]
    w["t_rc"] = batch["win_rc"]
    return w


def pair(batch, i):
    """(q bytes, t bytes, blocks ndarray) of pair i."""
    qo, ql = int(batch["q_off"][i]), int(batch["q_len"][i])
    to, tl = int(batch["t_off"][i]), int(batch["t_len"][i])
    bo, nb = int(batch["blk_off"][i]), int(batch["n_blk"][i])
    return (batch["q_arena"][qo:qo + ql], batch["t_arena"][to:to + tl], batch["blocks"][bo:bo + nb])
