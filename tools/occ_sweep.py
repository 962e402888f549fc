"""How the lane-per-pair fill scales with the waves resident per SIMD.  Substitution-only reads keep every band row 21
cells wide, so every pair is of the 32-slot class (9 KB of LDS per wave, seventeen waves per CU) and LDS no longer decides
the occupancy; the batch runs as ONE part of 1, 2, 3 and 4 waves per SIMD.  Run with GBA_LIB pointing at a build whose
fill kernel claims few enough registers (GBA_FILL_CLAIM) for the occupancy asked for."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GBA_PART_PAIRS"] = "4000000"
import torch
from blasr_amd import api, synth, _lib

ref = synth.reference(synth.ECOLI_LEN, synth.REF_SEED, 0.0)
for waves in (1, 2, 3, 4):
    n = 65536 * waves
    host = synth.generate(n, read_len=10000, p_ins=0.0, p_del=0.0, p_sub=0.15, seed=777, ref=ref)
    d = api.to_device(host)
    c = api.Context(0, timing=True)
    c.set_scoring(api.make_scoring(band=10))
    pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, needed = c.refine_device(d, pairs, None)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t = c.timing()
    ok = int((pairs.cpu().numpy().view(_lib.PAIR_RESULT_DT)["status"] == 0).sum())
    print("waves/SIMD %d pairs %d ok %d: call %.1f ms, fill %.2f ms, cells %.3e, fill GCUPS %.0f, wide pairs %d" % (
        waves, n, ok, dt * 1e3, t["fill_ms"], t["cells"], t["cells"] / t["fill_ms"] / 1e6, t["wide_pairs"]), flush=True)
    c.close()
    del d, pairs
    torch.cuda.empty_cache()
