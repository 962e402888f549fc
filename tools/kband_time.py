"""Throughput of gba_kband on three shapes: end extensions (500 x 500, band 10), the -global branch (10 kb, k 30,
quality-value profile) and the legacy affine refine (10 kb, k 36)."""
import sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blasr_amd import api, _lib
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kband_cases import pack, random_profile

rng = np.random.default_rng(1)
ACGT = np.frombuffer(b"ACGT", np.uint8)


def noisy(t, rate):
    r = rng.random(len(t))
    keep = r >= rate * 0.3
    q = t.copy()
    sub = (r >= rate * 0.3) & (r < rate * 0.5)
    q[sub] = rng.choice(ACGT, int(sub.sum()))
    q = q[keep]
    ins = rng.random(len(q)) < rate * 0.5
    return np.insert(q, np.flatnonzero(ins), rng.choice(ACGT, int(ins.sum())))


def run(name, n, length, k, P, profile=False, reps=3, fixed_k=False):
    pairs = []
    for _ in range(n):
        t = rng.choice(ACGT, length)
        pairs.append((noisy(t, 0.13), t))
    ks = [k if fixed_k else max(k, abs(len(q) - len(t))) for q, t in pairs]
    b = pack(pairs, ks, profile=random_profile(rng, pairs) if profile else None)
    ctx = api.Context(0)
    ctx.kband_host(b, P)
    t0 = time.perf_counter()
    for _ in range(reps):
        pr, bl, rc = ctx.kband_host(b, P)
    dt = (time.perf_counter() - t0) / reps
    cells = int(pr["ncells"].sum())
    print("%s: %d pairs, %.1f ms host to host, %.2f GCUPS, %.0f pairs/s, ok %d" % (name, n, dt * 1e3, cells / dt / 1e9, n / dt, int((pr["status"] == 0).sum())), flush=True)
    ctx.close()


run("extend 500x500 band 10", 65536, 500, 10, api.make_kband_params(align_type=_lib.KB_EXTEND, ins=4, dele=5), fixed_k=True)
run("global 10 kb k 30 profile", 4096, 10000, 30, api.make_kband_params(align_type=_lib.KB_GLOBAL), profile=True)
run("affine-hp 10 kb k 36", 4096, 10000, 36, api.make_kband_params(align_type=_lib.KB_GLOBAL, gap_model=_lib.KB_AFFINE_HP))
