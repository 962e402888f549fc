#!/usr/bin/env python3
"""Benchmark of the guided banded alignment path (BLASR refine stage) on MI355X.

One step = one pass of the whole hot path (guide rows, DP fill, traceback, block
emission) over one batch of synthetic pairs that is already resident in HBM.
Workload at N=1: BASELINE config 2 -- 10 kb reads, 15 % indel-heavy error, 4.6 Mb
reference, linear gaps, band 10 (--workload selects configs 3, 4, 5).  With N>1 there
is ONE common synthetic read set; rank r (one per GPU) owns the reads with index
= r (mod N) and all their candidates, generates exactly those, and after the timed
steps writes their results at the reads' own places into one node-wide array
(blasr_amd.shard.SharedMerge).  No data-path collective; the only communication is
the timing barrier / max-reduce.  --scaling weak (default): per-GPU work is fixed, the
common set grows with N; --scaling strong: the common set is fixed (--pairs in all) and
split N ways (BASELINE config 4: "reads sharded across 2/4/8").

`python bench.py --gpus N` without a launcher starts the N ranks itself (one process per
GPU through torch.distributed.run, before anything in this process touches a GPU) and
fails when fewer than N GPUs are visible; under the driver's own launcher (WORLD_SIZE set)
it is one of the ranks.

Beside the device-resident rate (`value`) the line carries, on rank 0 at N=1:
  host_to_host  the same batch through gba_submit / gba_wait from and to page-locked
                host memory (targets as windows of the resident reference)
  cpu_baseline  the CPU oracle on all host cores over a bounded sample of the batch
  other_workloads  BASELINE configs 3, 4 and 5 device-resident, two steps each; for the one-part batches (3, 5) also streamed
                through four contexts; the default read set with a quality-value cost profile, and under the Local end rule
  streamed      the same batch streamed through two contexts by two threads (device-resident)
  end_to_end    reads in, filtered and scored alignments out (read-level harness)
host_to_host and end_to_end are measured by processes of their own, which run first, before this process touches the
GPU (host_legs_first); everything else follows the timed steps.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# A context runs the parts of a batch on a dozen streams (three slots of a main and two auxiliary streams, two copy
# streams).  With HIP's default of four hardware queues streams share queues and one part's guide construction waits
# behind another part's fill; the library's callers are told to raise it, and so does this program,
# before the HIP runtime starts.  Twenty: two contexts' kernel streams then find queues of their own (the same batch
# streamed through two contexts 913-924 GCUPS against 880-883 with sixteen), and the process stays below the number
# from which everything slows down (24 and 32: the contexts made later in the process ran at 55-75 % of their rate).
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("GBA_BENCH_HW_QUEUES", "20")      # (the GPU boxes export the default, 4)
sys.path.insert(0, ROOT)

# HBM bytes of one step's fill launches at the default workload, from separate rocprofv3 --pmc passes of
# this command (profiles/, WRITE_SIZE + 2 x FETCH_SIZE: the gfx950 correction for streamed reads)
MEASURED_FILL_TRAFFIC = {("c2_10kb_linear", 491520): 125906804 * 1024 + 2 * 18229433 * 1024}
# The host legs (host to host, end to end) run in processes of their own BEFORE this process touches the GPU: an idle HIP
# context beside them (its hardware queues stay allocated) costs them a third of their rate (host to host 472 against
# 651 GCUPS, to intervals 569 against 844 on the same box).
HOST_HW_QUEUES = os.environ.get("GBA_BENCH_HOST_HW_QUEUES", "16")
OPS_PER_CELL = {"linear": 8, "affine": 18}         # algorithmic int32 ops per DP cell
BYTES_PER_CELL = {"linear": 0.25, "affine": 0.75}  # algorithmic traceback bytes per cell
VALU_PEAK_TOPS = 78.6       # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz int32 lane-ops/s (= half the 157.3 TFLOP/s fp32 vector peak)
HBM_PEAK_GBS = 8000.0
DEFAULT_PAIRS = {"c2_10kb_linear": 491520, "c3_20kb_affine": 16384, "c4_chr20_bestn10": 163840,
                 "c5_50kb_wide": 2048, "c1_ecoli_plumbing": 65536}


def host_cores():
    """Threads the CPU baseline may use: the CPUs this process may run on, capped by the
    container's CPU quota when there is one (the affinity mask of a shared host shows every CPU of
    the machine, the quota is what this job really gets)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        n = min(n, int(env))            # the box exports its CPU share this way
    return n


def cpu_baseline(batch, scoring_kw, seconds, threads):
    """Oracle (C restatement, 'port') on host cores over a time-bounded sample of the same pairs:
    pthread workers with private reusable work space, pairs handed out from a shared counter."""
    from oracle import oracle
    P = oracle.make_params(**scoring_kw)
    _, _, _, pairs_done, cells, dt = oracle.align_batch_mt(batch, P, threads, seconds)
    return cells / dt / 1e9, pairs_done, dt


def results_digest(pairs, blocks):
    """A few sums that two runs of the same batch must agree on."""
    return [int(pairs["score"].astype(np.int64).sum()), int(pairs["n_blk"].astype(np.int64).sum()),
            int((pairs["status"] == 0).sum()), int(np.asarray(blocks).view(np.uint32).astype(np.uint64).sum())]


def host_child(args):
    """The host-to-host leg, in a process of its own: no PyTorch (the system's HIP runtime moves both
    copy directions on the copy engines at once; the one bundled with PyTorch falls back to copy kernels
    that take the GPU from the DP kernels), GPU_MAX_HW_QUEUES=16, and two contexts used in turn on the
    GPU, each with its batch and result buffers in page-locked memory: what two of the reference's
    worker threads sharing one GPU would do."""
    from blasr_amd import api, synth, _lib
    cfg = dict(synth.PRESETS[args.workload])
    if args.band is not None:
        cfg["band"] = args.band
    affine = cfg["mode"] == "affine"
    k = cfg.get("decoy_every", 0) or 1
    n = args.pairs
    ref = synth.reference(cfg["ref_len"], synth.REF_SEED, cfg.get("n_frac", 0.0))
    host = synth.generate(n, share_reads=k > 1, ref=ref, **{a: b for a, b in cfg.items() if a != "ref"})
    host = synth.as_windows(synth.compact_blocks(host))
    pipe = api.HostPipeline(0, depth=args.host_contexts, stagger=float(os.environ.get("BENCH_STAGGER", "0.4")))
    pipe.set_scoring(api.make_scoring(band=cfg["band"], max_band=args.max_band, mode=_lib.MODE_AFFINE if affine else _lib.MODE_LINEAR))
    pipe.set_reference(ref)
    # room for the refined blocks: what the caller says the batch needs plus a twelfth, else a block per four bases
    cap_blocks = args.host_blocks + args.host_blocks // 12 if args.host_blocks else int(host["q_len"].astype(np.int64).sum() // 4) + n
    bufs = []
    for _ in range(args.host_contexts):
        bufs.append((pipe.pinned_copy(host), pipe.host_array(n, _lib.PAIR_RESULT_DT), pipe.host_array(cap_blocks, _lib.BLOCK_DT)))
    for i in range(args.host_contexts):           # warm-up: every context once (its device buffers grow to size)
        pipe.submit(*bufs[i]); rc, needed = pipe.wait()
        if rc != 0:
            raise SystemExit("host path rc=%d" % rc)
    t0 = time.perf_counter()
    pipe.submit(*bufs[0])
    for i in range(args.host_steps):
        if i + 1 < args.host_steps:
            pipe.submit(*bufs[(i + 1) % args.host_contexts])
        rc, needed = pipe.wait()
    dt = (time.perf_counter() - t0) / args.host_steps
    hb, pairs, blocks = bufs[(args.host_steps - 1) % args.host_contexts]
    dig_full = results_digest(pairs, blocks[:needed])
    # the same without a block arena (interval / summary output): only the pair results come back
    for b in bufs:
        b[1]["score"][:] = 0
    t1 = time.perf_counter()
    pipe.submit(bufs[0][0], bufs[0][1], None)
    for i in range(args.host_steps):
        if i + 1 < args.host_steps:
            pipe.submit(bufs[(i + 1) % args.host_contexts][0], bufs[(i + 1) % args.host_contexts][1], None)
        rc2, needed2 = pipe.wait()
    dt_iv = (time.perf_counter() - t1) / args.host_steps
    iv_same = rc2 == 0 and needed2 == needed and results_digest(pairs, blocks[:0])[:3] == dig_full[:3]
    h2d = sum(hb[x].nbytes for x in ("q_arena", "blocks", "q_off", "t_off", "blk_off", "q_len", "t_len", "n_blk", "t_rc"))
    # the same batches with their reads 2 bits per base (gba_batch.q_packed): a quarter of the read bytes cross the link
    pk = api.pack_reads_2bit(host)
    pbufs = []
    for b in bufs:
        qa = pipe.host_array(len(pk["q_arena"]), np.uint8)
        qa[...] = pk["q_arena"]
        pbufs.append(dict(b[0], q_arena=qa, q_bases=pk["q_bases"], q_packed=1))
    del pk
    for b in bufs:
        b[1]["score"][:] = 0
    t1 = time.perf_counter()
    pipe.submit(pbufs[0], bufs[0][1], bufs[0][2])
    for i in range(args.host_steps):
        if i + 1 < args.host_steps:
            j = (i + 1) % args.host_contexts
            pipe.submit(pbufs[j], bufs[j][1], bufs[j][2])
        rc4, needed4 = pipe.wait()
    dt_pk = (time.perf_counter() - t1) / args.host_steps
    pk_same = rc4 == 0 and results_digest(pairs, blocks[:needed4]) == dig_full
    h2d_pk = h2d - hb["q_arena"].nbytes + pbufs[0]["q_arena"].nbytes
    cells_guided = int(pairs["ncells"][pairs["status"] == 0].sum())
    # from sequences alone: the batches without their guide blocks, the guides made on the device in the same call
    pipe.set_guide_producer(api.make_sdp_params(max_fragments=1024 if cfg["read_len"] <= 25_000 else 4096))
    seqs = [{k: v for k, v in b[0].items() if k not in ("blocks", "blk_off", "n_blk")} for b in bufs]
    from_seq = {}
    for what, want_blocks in (("blocks", True), ("intervals", False)):
        for i in range(args.host_contexts):       # warm-up: the guide producer's scratch
            pipe.submit(seqs[i], bufs[i][1], bufs[i][2] if want_blocks else None); rc3, needed3 = pipe.wait()
        t2 = time.perf_counter()
        pipe.submit(seqs[0], bufs[0][1], bufs[0][2] if want_blocks else None)
        for i in range(args.host_steps):
            if i + 1 < args.host_steps:
                j = (i + 1) % args.host_contexts
                pipe.submit(seqs[j], bufs[j][1], bufs[j][2] if want_blocks else None)
            rc3, needed3 = pipe.wait()
        ok3 = pairs["status"] == 0
        from_seq[what] = {"ms_per_step": (time.perf_counter() - t2) / args.host_steps * 1e3, "rc": rc3, "pairs_ok": int(ok3.sum()),
                          "cells": int(pairs["ncells"][ok3].sum()), "blocks_out": int(needed3),
                          "score_per_base": float(pairs["score"][ok3].sum()) / max(1, int(pairs["rows"][ok3].sum()))}
    h2d_seq = sum(seqs[0][x].nbytes for x in seqs[0] if x != "n_pairs")
    print(json.dumps({"ms_per_step": dt * 1e3, "h2d_bytes": h2d, "d2h_bytes": n * _lib.PAIR_RESULT_DT.itemsize + needed * _lib.BLOCK_DT.itemsize,
                      "cells": cells_guided, "digest": dig_full, "from_sequences": from_seq, "h2d_bytes_sequences": h2d_seq,
                      "intervals_ms_per_step": dt_iv * 1e3, "intervals_ok": bool(iv_same),
                      "packed_ms_per_step": dt_pk * 1e3, "packed_h2d_bytes": h2d_pk, "packed_ok": bool(pk_same)}), flush=True)
    pipe.close()


def fixed_band(api, _lib, host, n, threads):
    """The aligner of the -extend step on the batch's own sequences: the last 500 bases of every read against the last
    500 of its window, backwards from their ends (ExtendAlignmentReverse's shape: band 10, drop-off 10), through
    gba_kband from host memory to host memory; then the CPU restatement of the same on `threads` cores."""
    m = min(n, 65536)
    ql = np.minimum(host["q_len"][:m], 500).astype(np.uint32); tl = np.minimum(host["t_len"][:m], 500).astype(np.uint32)
    qs = (host["q_off"][:m] + host["q_len"][:m] - ql).astype(np.int64); ts = (host["t_off"][:m] + host["t_len"][:m] - tl).astype(np.int64)
    q_off = np.zeros(m, np.uint64); t_off = np.zeros(m, np.uint64)
    q_off[1:] = np.cumsum(ql[:-1].astype(np.uint64)); t_off[1:] = np.cumsum(tl[:-1].astype(np.uint64))
    b = {"n_pairs": m, "q_arena": np.concatenate([host["q_arena"][qs[i]:qs[i] + ql[i]] for i in range(m)]),
         "t_arena": np.concatenate([host["t_arena"][ts[i]:ts[i] + tl[i]] for i in range(m)]),
         "q_off": q_off, "q_len": ql, "t_off": t_off, "t_len": tl, "k": np.full(m, 10, np.int32), "reverse": np.ones(m, np.uint8)}
    P = api.make_kband_params(align_type=_lib.KB_EXTEND, ins=4, dele=5, max_drops=10)
    ctx = api.Context(0)
    ctx.kband_host(b, P)
    t0 = time.perf_counter()
    pairs, blocks, rc = ctx.kband_host(b, P)
    dt = time.perf_counter() - t0
    ctx.close()
    ok = pairs["status"] == 0
    r = {"what": "gba_kband, GBA_KB_EXTEND backwards from the ends of the batch's reads and windows (500 x 500, band 10, drop-off 10), host to host",
         "pairs": m, "ms": round(dt * 1e3, 3), "extensions_per_sec": round(m / dt, 1), "gcups": round(int(pairs["ncells"][ok].sum()) / dt / 1e9, 2),
         "pairs_ok": int(ok.sum()), "extended": int((pairs["dp_score"][ok] < 0).sum()), "rc": rc}
    if threads:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as _o

        def one(i):
            q = b["q_arena"][int(q_off[i]):int(q_off[i]) + int(ql[i])]
            t = b["t_arena"][int(t_off[i]):int(t_off[i]) + int(tl[i])]
            return _o.kband_align(q, t, 10, P, reverse=True)[0]
        done, t1, cells, same = 0, time.perf_counter(), 0, True
        with ThreadPoolExecutor(threads) as pool:
            while time.perf_counter() - t1 < 3.0 and done + 64 * threads <= m:
                for i, w in zip(range(done, done + 64 * threads), pool.map(one, range(done, done + 64 * threads))):
                    cells += w["ncells"]
                    same = same and (w["dp_score"], w["score"], w["n_blk"]) == (int(pairs[i]["dp_score"]), int(pairs[i]["score"]), int(pairs[i]["n_blk"]))
                done += 64 * threads
        secs = time.perf_counter() - t1
        r["cpu_baseline"] = {"extensions_per_sec": round(done / secs, 1), "gcups": round(cells / secs / 1e9, 3), "cores": threads, "kind": "port",
                             "same_results": same, "sample": "%d pairs in %.1f s (oracle/kband_oracle.c through ctypes, %d threads)" % (done, secs, threads)}
    return r


def host_legs_first(args):
    """The legs that run in processes of their own -- the host-to-host child and the read-level harness -- before this
    process has touched the GPU; what they printed is folded into the line later.  A HIP context that sits idle beside
    them keeps its hardware queues, and the two contexts of a host pipeline then share what is left."""
    import subprocess
    from blasr_amd import synth
    r = {}
    if not args.no_host:
        k = dict(synth.PRESETS[args.workload]).get("decoy_every", 0) or 1
        n = args.pairs if args.pairs is not None else DEFAULT_PAIRS[args.workload]
        n -= n % k
        env = dict(os.environ, GBA_NO_TORCH="1", GPU_MAX_HW_QUEUES=HOST_HW_QUEUES)
        cmd = [sys.executable, os.path.abspath(__file__), "--host-child", "--workload", args.workload, "--pairs", str(n),
               "--max-band", str(args.max_band), "--host-steps", str(args.host_steps), "--host-contexts", str(args.host_contexts),
               "--host-blocks", "0"] + (["--band", str(args.band)] if args.band is not None else [])
        r["host_child"] = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if not args.no_end_to_end and args.workload == "c2_10kb_linear":
        from blasr_amd import build
        e = {"what": "read_harness: candidates from the generator -> gba_submit from sequences (guides made on the device, refine, "
                     "pair results only) -> per read sort / filters / mapQV / m4 text; two contexts in turn, 4 batches back to back"}
        env = dict(os.environ, GPU_MAX_HW_QUEUES=HOST_HW_QUEUES)
        for key, argv in (("c2_10kb_1_candidate", ["--reads", "122880", "--cands", "1", "--read-len", "10000", "--ref-len", str(synth.ECOLI_LEN), "--n-frac", "0", "--seed", "2000"]),
                          ("c4_chr20_10_candidates", ["--reads", "16384", "--cands", "10", "--read-len", "10000", "--ref-len", str(synth.CHR20_LEN), "--n-frac", "0.01", "--seed", "4000"])):
            try:
                p = subprocess.run([build.READ_HARNESS] + argv + ["--batches", "4"], env=env, capture_output=True, text=True, timeout=600)
                e[key] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else {"error": (p.stderr or p.stdout)[-300:]}
            except Exception as x:
                e[key] = {"error": repr(x)[:300]}
        r["end_to_end"] = e
    return r


def launch_ranks(n_gpus):
    """`bench.py --gpus N` started by hand: become the launcher of N ranks of this same command.  Nothing here touches
    a GPU (counting devices does not initialise one), so the ranks are started as children, never by replacing this
    process.  GBA_BENCH_BACKEND=gloo: a rehearsal, several ranks may share a GPU."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n_gpus and os.environ.get("GBA_BENCH_BACKEND", "nccl") != "gloo":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (GBA_BENCH_BACKEND=gloo rehearses with ranks sharing a GPU)" % (n_gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def kernel_name(cfg, tm, n, quad_min, qv=False, local=False):
    """The fill kernel that ran the bulk of a batch, from the library's own routing."""
    affine = cfg["mode"] == "affine"
    if qv or local:
        if tm["wide_pairs"] * 2 > n:
            return "k_fill_wide_affine<.., %s>" % str(qv).lower() if affine else "k_fill_wide<.., %s, %s>" % (str(qv).lower(), str(local).lower())
        return "k_fill<%s, %s, %s>" % (str(affine).lower(), str(qv).lower(), str(local).lower())
    if tm["wide_pairs"] * 2 > n:               # most pairs ran wave-per-pair
        if affine:
            return "k_fill_quad_affine" if tm["wide_pairs"] >= quad_min and cfg["band"] <= 30 else "k_fill_wide_affine"
        return "k_fill_wide_cols" if cfg["band"] > 30 else "k_fill_wide<1>"
    return "k_fill<%s, false, false>" % ("true" if affine else "false")


def stream_batches(torch, pipe, dbatch, bufs, warm, count):
    """`count` runs of the batch through the pipeline's contexts in turn, as many in flight as it has contexts, after
    `warm` untimed ones (scratch allocation).  Returns (seconds per batch, first non-zero return code or 0)."""
    depth = len(pipe.ctxs)
    rc = 0
    for phase, total in ((0, warm), (1, count)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(total):
            if k >= depth:
                rc = rc or pipe.wait()[0]
            pipe.submit(dbatch, *bufs[k % depth])
        for _ in range(min(depth, total)):
            rc = rc or pipe.wait()[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / count, int(rc)


def other_workload(torch, api, synth, _lib, name, n, steps, max_band, band=None, stream_depth=0, qv=False, local=False):
    """One of the other BASELINE configurations, device-resident like the headline: generate, one warm-up step, `steps`
    timed steps; the same definitions of GCUPS and of the fill's share of the int32 VALU peak.  stream_depth: also the
    rate of a caller that streams such batches through that many contexts, a worker thread each (the reference's worker
    threads, one MappingBuffers each, /root/reference/iblasr/Blasr.cpp:771): a batch of this size is one part, whose
    traceback then runs under the other contexts' fills, and the wave-per-pair fills of several batches share the SIMDs.
    qv: the batch carries a cost profile (gba_batch.q_profile, the quality-value scorers of
    /root/reference/iblasr/BlasrUtilsImpl.hpp:636-646): random costs made on the device, 11 per base (match -5, the rest
    1..29), so that every row of the DP looks its costs up in its own tables."""
    cfg = dict(synth.PRESETS[name])
    if band is not None:
        cfg["band"] = band
    affine = cfg["mode"] == "affine"
    k = cfg.get("decoy_every", 0) or 1
    n -= n % k
    ref = synth.reference(cfg["ref_len"], synth.REF_SEED, cfg.get("n_frac", 0.0))
    host = synth.generate(n, share_reads=k > 1, ref=ref, **{a: b for a, b in cfg.items() if a != "ref"})
    d = api.to_device(host, "cuda:0")
    cap = int(host["q_len"].astype(np.int64).sum() // 4) + n
    if qv:
        na = len(host["q_arena"])
        code = np.full(256, 4, np.int64)
        for i, c in enumerate(b"ACGT"):
            code[c] = i
        g = torch.Generator(device="cuda:0")
        g.manual_seed(1)
        prof = torch.randint(1, 30, (na, 11), dtype=torch.int16, device="cuda:0", generator=g)
        prof[torch.arange(na, device="cuda:0"), torch.from_numpy(code[host["q_arena"]]).to("cuda:0")] = -5
        d["q_profile"] = prof.reshape(-1)
        del prof
    del host
    out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
    out_blocks = torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0")
    ctx = api.Context(0, timing=True)
    # local: the Local end rule of the CCS call site (/root/reference/iblasr/BlasrAlignImpl.hpp:1418-1421) on the same pairs
    ctx.set_scoring(api.make_scoring(band=cfg["band"], max_band=max_band, mode=_lib.MODE_AFFINE if affine else _lib.MODE_LINEAR,
                                     align_type=_lib.ALIGN_LOCAL if local else _lib.ALIGN_GLOBAL))
    rc, _ = ctx.refine_device(d, out_pairs, out_blocks)
    torch.cuda.synchronize()
    fill = tb = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        rc, _ = ctx.refine_device(d, out_pairs, out_blocks)
        tm = ctx.timing()
        fill += tm["fill_busy_ms"]; tb += tm["traceback_ms"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT)
    cells = int(tm["cells"])
    fill_s = fill / steps / 1e3
    ops = OPS_PER_CELL[cfg["mode"]]
    r = {"pairs": n, "read_len": cfg["read_len"], "band": cfg["band"], "gaps": cfg["mode"], "candidates_per_read": k, "rc": rc,
         "pairs_ok": int((res["status"] == 0).sum()), "ms_per_step": round(dt * 1e3, 3), "gcups": round(cells / dt / 1e9, 2),
         "kernel": kernel_name(cfg, tm, n, int(os.environ.get("GBA_QUAD_AFFINE_MIN", "10240")), qv, local),
         "fill_ms": round(fill_s * 1e3, 3), "traceback_ms": round(tb / steps, 3), "fill_gcups": round(cells / fill_s / 1e9, 2),
         "frac": round(cells * ops / fill_s / 1e12 / VALU_PEAK_TOPS, 4), "steps": steps}
    if stream_depth > 1:
        pipe = api.DevicePipeline(0, depth=stream_depth, contexts=[ctx])
        pipe.set_scoring(api.make_scoring(band=cfg["band"], max_band=max_band, mode=_lib.MODE_AFFINE if affine else _lib.MODE_LINEAR))
        bufs = [(out_pairs, out_blocks)] + [(torch.zeros_like(out_pairs), torch.zeros_like(out_blocks)) for _ in range(stream_depth - 1)]
        count = stream_depth * max(steps, 4)
        dt2, rc2 = stream_batches(torch, pipe, d, bufs, stream_depth, count)
        r["streamed"] = {"what": "batches streamed through %d contexts, a worker thread each (api.DevicePipeline)" % stream_depth, "contexts": stream_depth,
                         "rc": rc2, "batches": count, "ms_per_batch": round(dt2 * 1e3, 3), "gcups": round(cells / dt2 / 1e9, 2),
                         "same_results": all(bool(torch.equal(bufs[0][0], b_[0])) for b_ in bufs[1:])}
        pipe.close()
        del bufs
    ctx.close()
    del d, out_pairs, out_blocks
    torch.cuda.empty_cache()
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=None, help="pairs per GPU per step (default: per workload)")
    ap.add_argument("--workload", default="c2_10kb_linear")
    ap.add_argument("--band", type=int, default=None, help="override the preset's band")
    ap.add_argument("--max-band", type=int, default=250, help="cap of the widened band in guide gaps (experiments)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-host", action="store_true", help="skip the host-to-host measurement")
    ap.add_argument("--no-guides", action="store_true", help="skip the guide-production measurement")
    ap.add_argument("--no-fixed-band", action="store_true", help="skip the gba_kband (end extension) measurement")
    ap.add_argument("--no-others", action="store_true", help="skip the other BASELINE configurations (other_workloads)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the read-level harness (end_to_end)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the two-context streaming measurement (streamed)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = --pairs per GPU (default), strong = --pairs in all, split over the GPUs")
    ap.add_argument("--cigar", type=int, default=None, help="also have the device write CIGAR operations (GBA_CIGAR_* flags); not part of the default line")
    ap.add_argument("--host-steps", type=int, default=6)
    ap.add_argument("--host-contexts", type=int, default=2)
    ap.add_argument("--host-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--host-blocks", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.host_child:
        return host_child(args)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))

    early = host_legs_first(args) if int(os.environ.get("WORLD_SIZE", "1")) == 1 else {}

    import torch
    from blasr_amd import api, shard, synth, _lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d started as one of %d ranks: the launcher's --nproc-per-node and --gpus must agree" % (args.gpus, world))
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
    n = args.pairs if args.pairs is not None else DEFAULT_PAIRS[args.workload]
    if args.band is not None:
        cfg["band"] = args.band
    mode = cfg["mode"]
    affine = mode == "affine"
    k = cfg.get("decoy_every", 0) or 1
    if args.scaling == "strong" and world > 1:
        n //= world                              # one fixed read set, a share per GPU
    n -= n % k                                   # whole reads
    scoring_kw = dict(band=cfg["band"], max_band=args.max_band, mode=_lib.MODE_AFFINE if affine else _lib.MODE_LINEAR)
    scoring = api.make_scoring(**scoring_kw)
    # this rank's reads of the common set (index = rank mod world), all their candidates with them
    ref = synth.reference(cfg["ref_len"], synth.REF_SEED, cfg.get("n_frac", 0.0))
    host = synth.generate(n, shard=(rank, world), share_reads=k > 1, ref=ref, **{a: b for a, b in cfg.items() if a != "ref"})
    dbatch = api.to_device(host, dev)
    cap = int(host["q_len"].astype(np.int64).sum() // 4) + n      # ~1 block per 7.5 bases on these workloads
    out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device=dev)
    out_blocks = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
    ctx = api.Context(local, timing=True)
    ctx.set_scoring(scoring)
    cigar = None
    if args.cigar is not None:
        cigar = (torch.zeros(2 * cap, dtype=torch.int32, device=dev), torch.zeros(n + 1, dtype=torch.int64, device=dev), args.cigar)

    needed_dev = [0]

    fill_sum = [0.0]

    def run_steps(count):
        fill = tb = 0.0
        fill_sum[0] = 0.0
        for _ in range(count):
            rc, needed = ctx.refine_device(dbatch, out_pairs, out_blocks, cigar=cigar)   # returns when the batch has completed
            if rc != 0:
                raise RuntimeError("refine_device rc=%d needed=%d" % (rc, needed))
            needed_dev[0] = needed
            tm = ctx.timing()
            fill += tm["fill_busy_ms"]; tb += tm["traceback_ms"]
            fill_sum[0] += tm["fill_ms"]
        return fill, tb

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    fill_ms, tb_ms = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    res = out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT)
    tm = ctx.timing()
    cells_step, rec_bytes = int(tm["cells"]), int(tm["record_bytes"])
    # The fill kernel with the GPU to itself: ONE part of the batch (what the library cuts large batches into), so no
    # other part's guide construction, traceback or fill runs beside its launches; after the timed region, not in `value`.
    alone = None
    part = 122880
    if rank == 0 and n > part + part // 15 and cigar is None:
        sub = dict(dbatch, n_pairs=part)
        for k_ in ("q_off", "q_len", "t_off", "t_len", "blk_off", "n_blk", "t_rc"):
            if sub.get(k_) is not None:
                sub[k_] = sub[k_][:part]
        for _ in range(2):
            rc_a, _needed = ctx.refine_device(sub, out_pairs, out_blocks)
        ta = ctx.timing()
        if rc_a == 0 and ta["fill_ms"] > 0:
            alone = {"pairs": part, "fill_ms": round(float(ta["fill_ms"]), 3), "cells": int(ta["cells"]),
                     "achieved": round(int(ta["cells"]) * OPS_PER_CELL[mode] / (ta["fill_ms"] * 1e-3) / 1e12, 3),
                     "frac": round(int(ta["cells"]) * OPS_PER_CELL[mode] / (ta["fill_ms"] * 1e-3) / 1e12 / VALU_PEAK_TOPS, 4),
                     "what": "the same kernel on one part of the batch run alone (in the step its launches share the GPU with the other parts' guide construction, tracebacks and fills)"}
    n_ok = int((res["status"] == 0).sum())
    cells_all, ok_all = cells_step, n_ok
    if world > 1:
        # every rank's pair results go to their places in ONE array in input order; rank 0 counts there
        name = "bench_%s" % os.environ.get("MASTER_PORT", "0")
        merge = None
        if rank == 0:
            merge = shard.SharedMerge(name, n * world, 0, create=True)
        dist.barrier()
        if rank != 0:
            merge = shard.SharedMerge(name, n * world, rank, create=False)
        merge.write(np.asarray(host["pair_id"]), res)
        dist.barrier()
        if rank == 0:
            allp = merge.pairs
            ok_all = int((allp["status"] == 0).sum())
            cells_all = int(allp["ncells"][allp["status"] == 0].sum())
        dist.barrier()
        merge.close()
        t = torch.tensor([cells_all, ok_all], dtype=torch.int64, device=red)
        dist.broadcast(t, 0)
        cells_all, ok_all = int(t[0].item()), int(t[1].item())
    total_cells = cells_all * args.steps
    gcups = total_cells / dt / 1e9
    fill_s = fill_ms / 1e3 / args.steps
    ops_cell, bytes_cell = OPS_PER_CELL[mode], BYTES_PER_CELL[mode]
    traffic = MEASURED_FILL_TRAFFIC.get((args.workload, n))
    kernel = kernel_name(cfg, tm, n, int(os.environ.get("GBA_QUAD_AFFINE_MIN", "10240")))      # the library's switch to four pairs per wave
    out = {
        "metric": "banded-DP GCUPS (guided banded alignment, refine stage)",
        "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": args.workload, "pairs_per_gpu": n, "read_len": cfg["read_len"],
                   "error": "ins %.2f / del %.2f / sub %.2f" % (cfg["p_ins"], cfg["p_del"], cfg["p_sub"]),
                   "candidates_per_read": k, "band": cfg["band"], "gaps": mode, "reference_len": cfg["ref_len"],
                   "cells_per_step_per_gpu": cells_step, "pairs_ok": ok_all, "pairs_total": n * world,
                   "parallelism": "one common read set of %d pairs, reads i = rank (mod %d) per GPU, no collective, results merged in input order" % (n * world, world)},
        "aligned_reads_per_sec": round(n * args.steps * world / dt, 1),
        "device_memory": {"ctx_bytes": int(tm["device_bytes"]), "part_pipeline_bytes": int(tm["slot_bytes"]),
                          "scratch_retries": int(tm["scratch_retries"]),
                          "what": "device memory the context holds after the timed steps (inputs and outputs are the caller's)"},
        "roofline": {
            "bound": "valu-int32", "kernel": kernel,
            "achieved": round(cells_step * ops_cell / fill_s / 1e12, 3), "peak": VALU_PEAK_TOPS,
            "unit": "Tops/s", "frac": round(cells_step * ops_cell / fill_s / 1e12 / VALU_PEAK_TOPS, 4),
            # fill_ms: the time of a step with at least one bulk fill launch in flight (HIP events around each part's launch on
            # its stream, the union of the intervals: consecutive parts' launches overlap, the next one's waves taking the SIMDs
            # the retiring one's leave); fill_launch_sum_ms counts such a stretch once per launch
            "ops_per_cell": ops_cell, "fill_ms": round(fill_s * 1e3, 3), "fill_launch_sum_ms": round(fill_sum[0] / args.steps, 3),
            "fill_gcups": round(cells_step / fill_s / 1e9, 2),
            "traffic": traffic,
            "alone": alone,
            # what inline-asm chains of single instructions reach on the box (tools/exp/valu.hip, two waves per SIMD and more):
            # the nominal peak above is not reachable by the instruction mix of the recurrence
            "measured_issue_rates_tops": {"v_add_u32 / v_and_b32 / v_mov_b32": 64.0, "v_min3_i32 / v_perm_b32 / v_lshl_add_u32 / v_bfi_b32 / v_bfe_u32": 36.0,
                                          "frac_of_v_add_rate": round(cells_step * ops_cell / fill_s / 1e12 / 64.0, 4)},
        },
        "roofline_hbm": {
            "bound": "hbm", "kernel": kernel,
            "achieved": round(cells_step * bytes_cell / fill_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(cells_step * bytes_cell / fill_s / 1e9 / HBM_PEAK_GBS, 5),
            "record_bytes_written": rec_bytes, "traffic": traffic,
        },
        "traceback_ms": round(tb_ms / args.steps, 3),
    }
    if cigar is not None:
        out["cigar"] = {"flags": args.cigar, "operations_per_step": ctx.cigar_needed}

    if rank == 0 and world == 1 and not args.no_streamed and cigar is None:
        # The same batches streamed: a second context on the GPU and a thread for each (the reference's worker threads with
        # a MappingBuffers each, /root/reference/iblasr/Blasr.cpp:771), so that one batch's first packing and guide and its
        # last traceback run under the other batch's fills.  After the timed region, not in `value`.
        try:
            pipe = api.DevicePipeline(local, depth=2, contexts=[ctx])
            pipe.set_scoring(scoring)
            bufs = [(out_pairs, out_blocks), (torch.zeros_like(out_pairs), torch.zeros_like(out_blocks))]
            count = 2 * max(args.steps, 4)
            dt2, rc2 = stream_batches(torch, pipe, dbatch, bufs, 2, count)
            out["streamed"] = {"what": "the same batch streamed through two contexts, a worker thread each (api.DevicePipeline), %d batches" % count,
                               "contexts": 2, "rc": rc2, "ms_per_batch": round(dt2 * 1e3, 3),
                               "gcups": round(cells_step / dt2 / 1e9, 2), "same_results": bool(torch.equal(bufs[0][0], bufs[1][0])),
                               "device_bytes": 2 * int(tm["device_bytes"])}
            pipe.close()
            del bufs, pipe
            torch.cuda.empty_cache()
        except Exception as e:              # a failure here must not cost the headline line
            out["streamed"] = {"error": repr(e)[:300]}

    if rank == 0 and world == 1 and not args.no_guides:
        # The step in front of the path, on the same batch: guides made on the device from the sequences
        # alone (k-mer matches chained, gba_sdp_device), then the refine stage run on those guides.
        sp = api.make_sdp_params(max_fragments=1024 if cfg["read_len"] <= 25_000 else 4096)      # ~31 maximal matches per kb at 15 % error
        g = ctx.sdp_device(dbatch, sp, blocks_capacity=cap)             # warm-up (scratch allocation)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        g = ctx.sdp_device(dbatch, sp, blocks_capacity=cap)
        torch.cuda.synchronize()
        sdp_s = time.perf_counter() - t1
        gst = g["status"].cpu().numpy()
        d2 = dict(dbatch, blocks=g["blocks"], blk_off=g["blk_off"], n_blk=g["n_blk"])
        ctx.refine_device(d2, out_pairs, out_blocks)
        t1 = time.perf_counter()
        rc2, _ = ctx.refine_device(d2, out_pairs, out_blocks)
        torch.cuda.synchronize()
        ref_s = time.perf_counter() - t1
        r2 = out_pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT)
        ok2 = r2["status"] == 0
        cells2 = int(r2["ncells"][ok2].sum())
        both = ok2 & (res["status"] == 0)
        out["guide_production"] = {
            "what": "gba_sdp_device: maximal exact matches of >= %d bases chained per pair (sdpIns %d, sdpDel %d), device-resident" % (sp.tuple_size, sp.ins, sp.dele),
            "ms_per_step": round(sdp_s * 1e3, 3), "pairs_per_sec": round(n / sdp_s, 1),
            "pairs_with_guide": int((gst == 0).sum()), "blocks": int(g["blocks_needed"]),
            "refine_on_these_guides": {"rc": rc2, "ms_per_step": round(ref_s * 1e3, 3), "gcups": round(cells2 / ref_s / 1e9, 2), "pairs_ok": int(ok2.sum()),
                                       "score_per_base": round(float(r2["score"][both].sum()) / max(1, int(r2["rows"][both].sum())), 4),
                                       "score_per_base_generator_guides": round(float(res["score"][both].sum()) / max(1, int(res["rows"][both].sum())), 4)},
            "guides_then_refine_gcups": round(cells2 / (sdp_s + ref_s) / 1e9, 2)}
        del g, d2
        # what is read below is the result of the generator's guides again
        ctx.refine_device(dbatch, out_pairs, out_blocks)
        torch.cuda.synchronize()

    # The headline context goes before the other configurations make theirs: its eleven streams would keep their places on
    # the hardware queues, and a later context's streams then share queues among themselves (a part's second fill launch
    # queued behind the previous part's traceback: C4 572-583 GCUPS instead of 712-719).
    digest = results_digest(res, out_blocks[:needed_dev[0]].cpu().numpy()) if rank == 0 and world == 1 and not args.no_host else None
    del dbatch, out_blocks, out_pairs
    ctx.close()
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_host:
        # The same batch from host memory to host memory (gba_submit / gba_wait), measured by a child process before this
        # one touched the GPU: see host_child() and host_legs_first().
        p = early["host_child"]
        if p.returncode != 0:
            out["host_to_host"] = {"error": (p.stderr or p.stdout)[-400:]}
        else:
            h = json.loads(p.stdout.strip().splitlines()[-1])
            hg = h["cells"] / (h["ms_per_step"] / 1e3) / 1e9
            out["host_to_host"] = {"gcups": round(hg, 2), "ms_per_step": round(h["ms_per_step"], 3),
                                   "h2d_gb": round(h["h2d_bytes"] / 1e9, 3), "d2h_gb": round(h["d2h_bytes"] / 1e9, 3),
                                   "frac_of_device_resident": round(hg / gcups, 3),
                                   "same_results_as_device_path": h["digest"] == digest,
                                   "intervals_only": {"gcups": round(h["cells"] / (h["intervals_ms_per_step"] / 1e3) / 1e9, 2),
                                                      "ms_per_step": round(h["intervals_ms_per_step"], 3),
                                                      "d2h_gb": round(n * 88 / 1e9, 3), "ok": h["intervals_ok"],
                                                      "what": "the same batches with gba_results.blocks == NULL: pair results (score, counts, "
                                                              "aligned interval) only, as interval (-m 4) output needs"},
                                   "packed_reads": {"gcups": round(h["cells"] / (h["packed_ms_per_step"] / 1e3) / 1e9, 2), "ms_per_step": round(h["packed_ms_per_step"], 3),
                                                    "h2d_gb": round(h["packed_h2d_bytes"] / 1e9, 3), "same_results": h["packed_ok"],
                                                    "what": "the same batches with the reads 2 bits per base (gba_batch.q_packed): a quarter of the read bytes cross the link and the kernels read the packed form"},
                                   "from_sequences": {
                                       "what": "the same batches WITHOUT their guide blocks (gba_set_guide_producer): reads and windows in, guides made "
                                               "on the device, refined alignments out; nothing of a guide crosses the link",
                                       "h2d_gb": round(h["h2d_bytes_sequences"] / 1e9, 3),
                                       **{k: {"ms_per_step": round(v["ms_per_step"], 3), "reads_per_sec": round(n / (v["ms_per_step"] / 1e3), 1),
                                              "gcups": round(v["cells"] / (v["ms_per_step"] / 1e3) / 1e9, 2), "rc": v["rc"], "pairs_ok": v["pairs_ok"],
                                              "blocks_out": v["blocks_out"], "score_per_base": round(v["score_per_base"], 4)}
                                          for k, v in h["from_sequences"].items()}},
                                   "how": "gba_submit / gba_wait, batch and results in page-locked memory (gba_host_alloc), targets = windows of "
                                          "the resident reference, %d contexts used in turn on the GPU, %d batches back to back, "
                                          "GPU_MAX_HW_QUEUES=%s, child process on the system HIP runtime" % (args.host_contexts, args.host_steps, HOST_HW_QUEUES)}

    if rank == 0 and world == 1 and not args.no_cpu:
        threads = host_cores()
        v, done, secs = cpu_baseline(host, dict(scoring_kw, mode=1 if affine else 0), args.cpu_seconds, threads)
        out["cpu_baseline"] = {"value": round(v, 4), "unit": "GCUPS", "cores": threads, "kind": "port",
                               "host_cpus_visible": len(os.sched_getaffinity(0)),
                               "sample": "%d pairs of the same batch in %.1f s (oracle: C restatement built -O3 -march=x86-64-v3, "
                                         "one thread per core with private work space; not the BLASR binary)" % (done, secs)}
        if "guide_production" in out:
            # the guide producer's CPU restatement on the same cores, a few seconds of the same batch
            from concurrent.futures import ThreadPoolExecutor
            from oracle import oracle as _o
            op = _o.make_sdp_params(k=sp.tuple_size, ins=sp.ins, dele=sp.dele, match=sp.match, max_fragments=sp.max_fragments)
            chunk, done, t1 = 8, 0, time.perf_counter()
            with ThreadPoolExecutor(threads) as pool:
                while time.perf_counter() - t1 < 4.0 and done + chunk * threads <= n:
                    list(pool.map(lambda f: _o.sdp_chain_batch(host, op, f, chunk), range(done, done + chunk * threads, chunk)))
                    done += chunk * threads
            secs = time.perf_counter() - t1
            out["guide_production"]["cpu_baseline"] = {"pairs_per_sec": round(done / secs, 1), "cores": threads, "kind": "port",
                                                       "sample": "%d pairs in %.1f s (oracle/sdp_oracle.c, one pair per call, %d threads)" % (done, secs, threads)}
    if rank == 0 and world == 1 and not args.no_fixed_band:
        out["fixed_band"] = fixed_band(api, _lib, host, n, None if args.no_cpu else host_cores())
    if rank == 0 and world == 1 and not args.no_others and args.workload == "c2_10kb_linear":
        # the other BASELINE configurations, device-resident, at the sizes the measurement plan names for them
        out["other_workloads"] = {}
        for key, name, pairs, band in (("c3_20kb_affine", "c3_20kb_affine", 16384, None), ("c5_50kb_band100", "c5_50kb_wide", 2048, 100),
                                       ("c4_chr20_bestn10", "c4_chr20_bestn10", 163840, None),
                                       ("c2_10kb_quality_values", "c2_10kb_linear", 122880, None),
                                       ("c2_10kb_local_end_rule", "c2_10kb_linear", 122880, None)):
            try:
                # (C4 is two parts, which overlap inside one context already: 2 % from a second one, not measured here)
                out["other_workloads"][key] = other_workload(torch, api, synth, _lib, name, pairs, 2, args.max_band, band,
                                                             stream_depth=4 if key in ("c3_20kb_affine", "c5_50kb_band100") else 0,
                                                             qv=key == "c2_10kb_quality_values", local=key == "c2_10kb_local_end_rule")
            except Exception as e:          # a failure here must not cost the headline line
                out["other_workloads"][key] = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_end_to_end and args.workload == "c2_10kb_linear":
        # Reads in, filtered and scored alignments out: the C++ read-level harness (blasr_amd/host/read_harness.cpp) runs the
        # candidates of every read through one from-sequences submit per batch and the reference's per-read steps behind
        # the refine stage (sort, RemoveLowQualityAlignments, RemoveOverlappingAlignments, StoreMapQVs, -m 4 lines) on host
        # threads while the next batch is on the GPU.  Seeding and clustering (where candidates come from) are not part of it.
        out["end_to_end"] = early["end_to_end"]
        e2 = out["end_to_end"].get("c2_10kb_1_candidate", {})
        if "reads_per_sec" in e2:
            out["end_to_end_reads_per_sec"] = e2["reads_per_sec"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
