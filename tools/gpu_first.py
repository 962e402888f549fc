"""First-light check on the GPU: small synthetic batches through the C-ABI vs the oracle."""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
from blasr_amd import synth, api
from oracle import oracle
import util

ctx = api.Context(0, timing=True)
ctx.set_scoring(api.make_scoring())
P = oracle.make_params()
for name, n, kw in (("tiny", 8, dict(read_len=300)), ("c1", 100, {}), ("c2", 130, {})):
    if name == "tiny":
        b = synth.generate(n, **kw)
    else:
        b, _ = synth.preset("c1_ecoli_plumbing" if name == "c1" else "c2_10kb_linear", n)
    t0 = time.time()
    pairs, blocks, rc = ctx.refine_host(b)
    dt = time.time() - t0
    print(name, "rc", rc, "statuses", np.bincount(pairs["status"]), "time", round(dt, 3), ctx.timing(), flush=True)
    util.compare_batch(oracle, b, P, pairs, blocks)
    print(name, "PARITY OK", flush=True)
