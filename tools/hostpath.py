"""The host-to-host path alone: gba_submit / gba_wait from and to page-locked memory, a few steps."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=491520)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--pageable", action="store_true")
ap.add_argument("--arena", action="store_true", help="send the targets as an arena instead of windows of the reference")
a = ap.parse_args()
if not os.environ.get('GBA_NO_TORCH'):
    import torch  # noqa: F401
from blasr_amd import api, synth, _lib
cfg = dict(synth.PRESETS["c2_10kb_linear"])
ref = synth.reference(cfg["ref_len"])
host = synth.compact_blocks(synth.generate(a.pairs, ref=ref, **cfg))
ctx = api.Context(0, timing=True)
ctx.set_scoring(api.make_scoring())
if not a.arena:
    ctx.set_reference(ref)
    host = synth.as_windows(host)
n = a.pairs
cap = int(host["q_len"].astype(np.int64).sum() // 6) + n
if a.pageable:
    hb = host
    pairs = np.zeros(n, _lib.PAIR_RESULT_DT); blocks = np.zeros(cap, _lib.BLOCK_DT)
else:
    hb = ctx.pinned_copy(host)
    pairs = ctx.host_array(n, _lib.PAIR_RESULT_DT); blocks = ctx.host_array(cap, _lib.BLOCK_DT)
for s in range(a.steps + 1):
    t0 = time.perf_counter()
    ctx.submit_host(hb, pairs, blocks)
    t1 = time.perf_counter()
    rc, needed = ctx.wait_host()
    t2 = time.perf_counter()
    cells = int(pairs["ncells"].sum())
    tm = ctx.timing()
    print("step %d rc %d submit %.2f ms total %.1f ms  %.1f GCUPS  fill_ms %.1f device total %.1f" % (s, rc, (t1 - t0) * 1e3, (t2 - t0) * 1e3, cells / (t2 - t0) / 1e9, tm["fill_ms"], tm["total_ms"]), flush=True)
ctx.close()
