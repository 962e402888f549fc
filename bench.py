#!/usr/bin/env python3
"""Benchmark of the guided banded alignment path (BLASR refine stage) on MI355X.

One step = one pass of the whole hot path (guide rows, DP fill, traceback, block
emission) over one batch of synthetic pairs that is already resident in HBM.
Workload at N=1: BASELINE config 2 -- 10 kb reads, 15 % indel-heavy error, 4.6 Mb
reference, linear gaps, band 10.  With N>1 every rank (one per GPU) owns an
independent shard of the reads (sharded by read index, no data-path collective);
the only communication is the timing barrier / max-reduce.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OPS_PER_CELL = 8            # algorithmic int32 ops per DP cell, linear gaps
BYTES_PER_CELL = 0.25       # algorithmic traceback bytes per cell (2-bit arrows)
VALU_PEAK_TOPS = 78.6       # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz int32 lane-ops/s (= half the 157.3 TFLOP/s fp32 vector peak)
HBM_PEAK_GBS = 8000.0


def cpu_baseline(batch, seconds, threads):
    """Oracle (C restatement, 'port') on host cores over a time-bounded sample of the same pairs."""
    from oracle import oracle
    P = oracle.make_params()
    n = int(batch["n_pairs"])
    chunk = 4
    nxt = [0]
    cells = [0]
    pairs_done = [0]
    lock = threading.Lock()
    deadline = time.time() + seconds

    def work():
        while time.time() < deadline:
            with lock:
                first = nxt[0]
                nxt[0] += chunk
            if first >= n:
                return
            res, _ = oracle.align_batch(batch, P, first, min(chunk, n - first))
            with lock:
                cells[0] += int(res["ncells"][res["status"] == 0].sum())
                pairs_done[0] += len(res)

    t0 = time.time()
    ths = [threading.Thread(target=work) for _ in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.time() - t0
    return cells[0] / dt / 1e9, pairs_done[0], dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=106496, help="pairs per GPU per step")
    ap.add_argument("--workload", default="c2_10kb_linear")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    from blasr_amd import api, synth, _lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the alignment path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local

    cfg = dict(synth.PRESETS[args.workload])
    # shard by read index: rank r owns the reads with global index = r (mod world); each rank
    # draws its own shard from a rank-specific seed, per-GPU work is fixed (weak scaling)
    cfg["seed"] = cfg["seed"] + 7919 * rank
    host = synth.generate(args.pairs, **cfg)
    scoring = api.make_scoring(band=cfg["band"])
    dbatch = api.to_device(host, dev)
    n = args.pairs
    cap = int(host["q_len"].astype(np.int64).sum() // 4) + n       # ~1 block per 7.5 bases on this workload
    out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device=dev)
    out_blocks = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
    ctx = api.Context(local, timing=True)
    ctx.set_scoring(scoring)

    def step():
        rc, needed = ctx.refine_device(dbatch, out_pairs, out_blocks)
        if rc != 0:
            raise RuntimeError("refine_device rc=%d needed=%d cap=%d" % (rc, needed, cap))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    fill_ms = tb_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()                       # gba_submit_device returns when the batch has completed
        tm = ctx.timing()
        fill_ms += tm["fill_ms"]
        tb_ms += tm["traceback_ms"]
    barrier()
    dt = time.perf_counter() - t0
    tm = ctx.timing()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: every pair aligned
    res = out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT)
    n_ok = int((res["status"] == 0).sum())
    cells_step = int(tm["cells"])
    total_cells = cells_step * args.steps * world
    gcups = total_cells / dt / 1e9
    fill_s = fill_ms / 1e3 / args.steps
    out = {
        "metric": "banded-DP GCUPS (guided banded alignment, refine stage)",
        "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": args.workload, "pairs_per_gpu": n, "read_len": cfg["read_len"],
                   "error": "ins %.2f / del %.2f / sub %.2f" % (cfg["p_ins"], cfg["p_del"], cfg["p_sub"]),
                   "band": cfg["band"], "gaps": cfg["mode"], "reference_len": cfg["ref_len"],
                   "cells_per_step_per_gpu": cells_step, "pairs_ok": n_ok,
                   "parallelism": "reads sharded by index, %d rank(s), no collective" % world},
        "aligned_reads_per_sec": round(n * args.steps * world / dt, 1),
        "roofline": {
            "bound": "valu-int32", "kernel": "k_fill_linear",
            "achieved": round(cells_step * OPS_PER_CELL / fill_s / 1e12, 3), "peak": VALU_PEAK_TOPS,
            "unit": "Tops/s", "frac": round(cells_step * OPS_PER_CELL / fill_s / 1e12 / VALU_PEAK_TOPS, 4),
            "ops_per_cell": OPS_PER_CELL, "fill_ms": round(fill_s * 1e3, 3),
            "fill_gcups": round(cells_step / fill_s / 1e9, 2),
            "traffic": None,
        },
        "roofline_hbm": {
            "bound": "hbm", "kernel": "k_fill_linear",
            "achieved": round(cells_step * BYTES_PER_CELL / fill_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(cells_step * BYTES_PER_CELL / fill_s / 1e9 / HBM_PEAK_GBS, 5),
            "record_bytes_written": int(tm["record_bytes"]), "traffic": None,
        },
        "traceback_ms": round(tb_ms / args.steps, 3),
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        threads = min(16, len(os.sched_getaffinity(0)))
        v, done, secs = cpu_baseline(host, args.cpu_seconds, threads)
        out["cpu_baseline"] = {"value": round(v, 4), "unit": "GCUPS", "cores": threads, "kind": "port",
                               "sample": "%d pairs of the same batch in %.1f s (oracle C restatement, not the BLASR binary)" % (done, secs)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
