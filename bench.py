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

# HBM bytes of one step's k_fill launches, from separate rocprofv3 --pmc passes of this command at the
# default workload (profiles/r01_step8_hbm_pmc.txt): WRITE_SIZE + 2 x FETCH_SIZE (gfx950 correction)
MEASURED_FILL_TRAFFIC = {("c2_10kb_linear", 491520): 162394798 * 1024 + 2 * 15918231 * 1024}
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
    ap.add_argument("--pairs", type=int, default=491520, help="pairs per GPU per step")
    ap.add_argument("--workload", default="c2_10kb_linear")
    ap.add_argument("--band", type=int, default=None, help="override the preset's band")
    ap.add_argument("--max-band", type=int, default=250, help="cap of the widened band in guide gaps (experiments)")
    ap.add_argument("--streams", type=int, default=1, help="contexts per GPU, each on its own stream with pairs/streams pairs")
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
        # GBA_BENCH_BACKEND=gloo: rehearsal of the multi-rank path with several ranks on one GPU
        backend = os.environ.get("GBA_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the alignment path has no CPU fallback")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    red = dev if world == 1 or dist.get_backend() == "nccl" else "cpu"      # where the timing reductions live

    cfg = dict(synth.PRESETS[args.workload])
    # shard by read index: rank r owns the reads with global index = r (mod world); each rank
    # draws its own shard from a rank-specific seed, per-GPU work is fixed (weak scaling)
    cfg["seed"] = cfg["seed"] + 7919 * rank
    S = max(1, args.streams)
    n = args.pairs
    per = [n // S + (1 if i < n % S else 0) for i in range(S)]
    hosts, dbatches, outs, ctxs = [], [], [], []
    if args.band is not None:
        cfg["band"] = args.band
    affine = cfg["mode"] == "affine"
    scoring = api.make_scoring(band=cfg["band"], max_band=args.max_band, mode=_lib.MODE_AFFINE if affine else _lib.MODE_LINEAR)
    for i in range(S):
        c2 = dict(cfg); c2["seed"] = cfg["seed"] + 104729 * i
        h = synth.generate(per[i], **c2)
        hosts.append(h)
        dbatches.append(api.to_device(h, dev))
        cap = int(h["q_len"].astype(np.int64).sum() // 4) + per[i]      # ~1 block per 7.5 bases on this workload
        outs.append((torch.zeros(per[i] * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device=dev),
                     torch.zeros((cap, 3), dtype=torch.int32, device=dev)))
        ctx = api.Context(local, timing=True)
        ctx.set_scoring(scoring)
        ctxs.append(ctx)
    host = hosts[0]

    def run_steps(i, k, acc):
        for _ in range(k):
            rc, needed = ctxs[i].refine_device(dbatches[i], outs[i][0], outs[i][1])   # returns when the batch has completed
            if rc != 0:
                raise RuntimeError("refine_device rc=%d needed=%d" % (rc, needed))
            tm = ctxs[i].timing()
            acc[0] += tm["fill_ms"]; acc[1] += tm["traceback_ms"]

    def run_all(k):
        accs = [[0.0, 0.0] for _ in range(S)]
        if S == 1:
            run_steps(0, k, accs[0])
        else:
            ths = [threading.Thread(target=run_steps, args=(i, k, accs[i])) for i in range(S)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        return accs

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_all(args.warmup)
    barrier()
    t0 = time.perf_counter()
    accs = run_all(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    fill_ms = sum(a[0] for a in accs) / S
    tb_ms = sum(a[1] for a in accs) / S
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: every pair aligned
    n_ok = 0
    cells_step = 0
    rec_bytes = 0
    for i in range(S):
        res = outs[i][0].cpu().numpy().view(_lib.PAIR_RESULT_DT)
        n_ok += int((res["status"] == 0).sum())
        tm = ctxs[i].timing()
        cells_step += int(tm["cells"])
        rec_bytes += int(tm["record_bytes"])
    cells_all, ok_all = cells_step, n_ok
    if world > 1:        # every rank has its own reads: sum the real counts
        t = torch.tensor([cells_step, n_ok], dtype=torch.int64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        cells_all, ok_all = int(t[0].item()), int(t[1].item())
    total_cells = cells_all * args.steps
    gcups = total_cells / dt / 1e9
    fill_s = fill_ms / 1e3 / args.steps
    ops_cell, bytes_cell = (18, 0.75) if affine else (OPS_PER_CELL, BYTES_PER_CELL)   # algorithmic work per cell
    traffic = MEASURED_FILL_TRAFFIC.get((args.workload, n)) if S == 1 else None
    out = {
        "metric": "banded-DP GCUPS (guided banded alignment, refine stage)",
        "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": args.workload, "pairs_per_gpu": n, "read_len": cfg["read_len"],
                   "error": "ins %.2f / del %.2f / sub %.2f" % (cfg["p_ins"], cfg["p_del"], cfg["p_sub"]),
                   "streams": S, "band": cfg["band"], "gaps": cfg["mode"], "reference_len": cfg["ref_len"],
                   "cells_per_step_per_gpu": cells_step, "pairs_ok": ok_all, "pairs_total": n * world,
                   "parallelism": "reads sharded by index, %d rank(s), no collective" % world},
        "aligned_reads_per_sec": round(n * args.steps * world / dt, 1),
        "roofline": {
            "bound": "valu-int32", "kernel": "k_fill<%s>" % ("true" if affine else "false"),
            "achieved": round(cells_step * ops_cell / fill_s / 1e12, 3), "peak": VALU_PEAK_TOPS,
            "unit": "Tops/s", "frac": round(cells_step * ops_cell / fill_s / 1e12 / VALU_PEAK_TOPS, 4),
            "ops_per_cell": ops_cell, "fill_ms": round(fill_s * 1e3, 3),     # bulk fill launches of a step, summed (HIP events)
            "fill_gcups": round(cells_step / fill_s / 1e9, 2),
            "traffic": traffic,
        },
        "roofline_hbm": {
            "bound": "hbm", "kernel": "k_fill<false>",
            "achieved": round(cells_step * bytes_cell / fill_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(cells_step * bytes_cell / fill_s / 1e9 / HBM_PEAK_GBS, 5),
            "record_bytes_written": rec_bytes, "traffic": traffic,
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
