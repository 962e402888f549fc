"""Host-to-host path with two contexts on one GPU used alternately: batch k+1 is submitted before batch k
is waited for, so its inputs travel while batch k runs and batch k's last results travel while it starts."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=491520)
ap.add_argument("--batches", type=int, default=8)
ap.add_argument("--ctxs", type=int, default=2)
a = ap.parse_args()
from blasr_amd import api, synth, _lib
cfg = dict(synth.PRESETS["c2_10kb_linear"])
ref = synth.reference(cfg["ref_len"])
host = synth.as_windows(synth.compact_blocks(synth.generate(a.pairs, ref=ref, **cfg)))
n = a.pairs
cap = int(host["q_len"].astype(np.int64).sum() // 6) + n
ctxs, bufs = [], []
for i in range(a.ctxs):
    c = api.Context(0)
    c.set_scoring(api.make_scoring()); c.set_reference(ref)
    ctxs.append(c)
    bufs.append((c.pinned_copy(host), c.host_array(n, _lib.PAIR_RESULT_DT), c.host_array(cap, _lib.BLOCK_DT)))
def submit(i):
    hb, p, b = bufs[i % a.ctxs]
    ctxs[i % a.ctxs].submit_host(hb, p, b)
# warm up every context once
for i in range(a.ctxs):
    submit(i); ctxs[i].wait_host()
t0 = time.perf_counter()
submit(0)
for k in range(a.batches):
    if a.ctxs == 1:                       # one context: one batch in flight
        if k > 0:
            submit(k)
    elif k + 1 < a.batches:
        submit(k + 1)
    rc, needed = ctxs[k % a.ctxs].wait_host()
    assert rc == 0
    print("batch %d done at %.1f ms" % (k, (time.perf_counter() - t0) * 1e3), flush=True)
dt = time.perf_counter() - t0
cells = int(bufs[0][1]["ncells"].sum())
print("%d ctxs: %.1f ms per batch, %.1f GCUPS" % (a.ctxs, dt / a.batches * 1e3, cells * a.batches / dt / 1e9))
