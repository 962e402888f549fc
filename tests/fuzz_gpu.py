#!/usr/bin/env python3
"""Randomised differential test: libgba (GPU, through the C ABI) against the oracle.

Draws random workloads (synthetic reads with random error rates and guide sparsity, and
adversarial pairs whose guides are random increasing blocks over random sequences) and random
scoring / band / mode parameters, runs both sides and compares every result field and every block.
Every second round also asks the device for the CIGAR operations (random flags) and compares a
sample of pairs with blasr_amd.emit.cigar_ops on the returned blocks; every third round runs the
guide producer (gba_sdp, random tuple size / penalties / fragment budget) on the round's sequences
against oracle/sdp_oracle.c; every third round runs the fixed-band aligners (gba_kband: random type, gap model, costs,
bands, strand flags, cost profile) on a sample of the round's sequences against oracle/kband_oracle.c; every fourth round
repeats the batch with the costs of a quality-value scorer (gba_batch.q_profile) against gbo_align_qv; every fifth round
(linear gaps) repeats it under the Local end rule (gba_scoring.align_type) against the oracle's; every fifth round repeats it
with the reads 2 bits per base and the targets as windows of a resident (packed) reference, on both strands, against the
same calls on letters and explicit targets.
Mismatches are written to gpurun_out/fuzz_fail_<k>.json with everything needed to replay them.

usage: python tests/fuzz_gpu.py [--seconds 120] [--seed 1]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

FIELDS = ("status", "dp_score", "score", "q_pos", "t_pos", "n_blk", "n_match", "n_mismatch",
          "n_ins", "n_del", "rows", "cols", "ncells")


def random_scoring(rng, _lib):
    mode = int(rng.choice([_lib.MODE_LINEAR, _lib.MODE_LINEAR, _lib.MODE_AFFINE, _lib.MODE_KBAND]))
    kw = dict(mode=mode, tie_break=int(rng.integers(0, 2)))
    lim = 40 if mode == _lib.MODE_AFFINE else 100
    kw["match"] = int(rng.integers(-lim, 3))
    kw["mismatch"] = int(rng.integers(0, lim))
    kw["n_score"] = int(rng.integers(-5, 8))
    kw["ins"] = int(rng.integers(0, lim))
    kw["dele"] = int(rng.integers(0, lim))
    kw["indel"] = int(rng.integers(0, 12))
    kw["band"] = int(rng.choice([0, 1, 2, 3, 5, 10, 16, 25, 40, 70, 120, 250, 300]))
    kw["max_band"] = int(rng.choice([5, 12, 30, 60, 250, 400]))
    if mode == _lib.MODE_AFFINE:
        kw["affine_open"] = int(rng.integers(0, 30))
        kw["affine_extend"] = int(rng.integers(0, 20))
    if mode == _lib.MODE_KBAND:
        kw["kband_k"] = int(rng.choice([0, 1, 4, 15, 60, 200, 500]))
    if rng.random() < 0.3:
        kw["stats_affine"] = 1
        kw["stats_affine_open"] = int(rng.integers(0, 12))
        kw["stats_affine_extend"] = int(rng.integers(0, 6))
    return kw


def random_synth(rng, synth):
    n = int(rng.integers(8, 160))
    read_len = int(rng.choice([40, 150, 600, 1500, 3000, 8000]))
    return synth.generate(
        n, read_len=read_len, len_jitter=int(rng.integers(0, read_len // 2 + 1)),
        p_ins=float(rng.uniform(0, 0.15)), p_del=float(rng.uniform(0, 0.12)), p_sub=float(rng.uniform(0, 0.08)),
        min_block=int(rng.integers(4, 20)), drop_frac=float(rng.uniform(0, 0.7)),
        seed=int(rng.integers(1, 1 << 30)), decoy_every=int(rng.choice([0, 0, 3, 10])),
        n_frac=float(rng.choice([0.0, 0.0, 0.05])), ref_len=200_000)


def random_adversarial(rng, util):
    """Pairs whose guide has nothing to do with the sequences: random increasing blocks."""
    pairs = []
    for _ in range(int(rng.integers(4, 64))):
        ql, tl = int(rng.integers(1, 900)), int(rng.integers(1, 900))
        q = rng.choice(list(b"ACGTN"), size=ql, p=[0.24, 0.24, 0.24, 0.24, 0.04]).astype(np.uint8).tobytes()
        t = rng.choice(list(b"ACGTacgtN"), size=tl).astype(np.uint8).tobytes()
        blocks, qp, tp = [], 0, 0
        for _b in range(int(rng.integers(0, 12))):
            qp += int(rng.integers(0, 60)); tp += int(rng.integers(0, 60))
            ln = int(rng.integers(1, 40))
            if qp + ln > ql or tp + ln > tl:
                break
            blocks.append((qp, tp, ln))
            qp += ln; tp += ln
        if rng.random() < 0.05 and len(blocks) > 1:
            blocks[0], blocks[1] = blocks[1], blocks[0]          # malformed guide
        pairs.append((q, t, blocks))
    return util.make_batch(pairs)


_LETTER = np.full(256, ord("N"), np.uint8)
for _c in b"ACGT":
    _LETTER[_c] = _c
    _LETTER[_c + 32] = _c


def check_cigars(rng, emit, _lib, batch, pairs, blocks, ops, off, flags):
    """A sample of the round's pairs: device operations == host restatement on the same blocks."""
    n = int(batch["n_pairs"])
    if int(off[n]) != len(ops):
        return (n, "cigar_total", int(off[n]), len(ops))
    for p in rng.choice(n, size=min(n, 12), replace=False):
        p = int(p)
        k = pairs[p]
        got = [(int(w) >> 4, _lib.CIGAR_OPS[int(w) & 15]) for w in ops[int(off[p]):int(off[p + 1])]]
        if int(k["status"]) != 0:
            if got:
                return (p, "cigar_of_failed_pair", 0, len(got))
            continue
        qo, ql = int(batch["q_off"][p]), int(batch["q_len"][p])
        to, tl = int(batch["t_off"][p]), int(batch["t_len"][p])
        q = _LETTER[batch["q_arena"][qo:qo + ql]].tobytes()
        t = _LETTER[batch["t_arena"][to:to + tl]].tobytes()
        kb = blocks[int(k["blk_off"]):int(k["blk_off"]) + int(k["n_blk"])]
        want = emit.cigar_ops(kb, q, t, int(k["q_pos"]), int(k["t_pos"]), use_seq_match=bool(flags & 1), allow_adjacent_indels=bool(flags & 2))
        if got != want:
            return (p, "cigar", len(want), len(got))
    return None


def check_sdp(rng, api, oracle, _lib, ctx, pool, batch):
    n = int(batch["n_pairs"])
    kw = dict(tuple_size=int(rng.integers(4, 16)), ins=int(rng.integers(0, 21)), dele=int(rng.integers(0, 21)),
              match=int(rng.integers(1, 11)), max_fragments=int(rng.choice([16, 64, 256, 1024, 4096])))
    st, sc, nb, off, blocks, rc = ctx.sdp_host(batch, api.make_sdp_params(**kw))
    if rc != 0:
        return (0, "sdp_rc", 0, rc), kw
    OP = oracle.make_sdp_params(k=kw["tuple_size"], ins=kw["ins"], dele=kw["dele"], match=kw["match"], max_fragments=kw["max_fragments"])
    chunk = max(1, n // 16)
    futs = [(f, pool.submit(oracle.sdp_chain_batch, batch, OP, f, min(chunk, n - f))) for f in range(0, n, chunk)]
    for f0, fu in futs:
        est, enb, esc, _, eb = fu.result()
        for i in range(len(est)):
            p = f0 + i
            if int(st[p]) == _lib.PAIR_TOO_LONG and int(est[i]) == _lib.PAIR_TOO_LONG:
                continue
            if (int(st[p]), int(nb[p]), int(sc[p])) != (int(est[i]), int(enb[i]), int(esc[i])):
                return (p, "sdp", (int(est[i]), int(enb[i]), int(esc[i])), (int(st[p]), int(nb[p]), int(sc[p]))), kw
            got = blocks[int(off[p]):int(off[p]) + int(nb[p])]
            if not (np.array_equal(got["q_pos"], eb[i]["q_pos"]) and np.array_equal(got["t_pos"], eb[i]["t_pos"]) and np.array_equal(got["length"], eb[i]["length"])):
                return (p, "sdp_blocks", len(eb[i]), len(got)), kw
    return None, kw


def check_kband(rng, api, oracle, _lib, ctx, pool, batch):
    """gba_kband with random type / gap model / costs / bands / strand flags / cost profile on (a sample of) the round's
    sequences against oracle/kband_oracle.c."""
    n_all = int(batch["n_pairs"])
    pick = rng.choice(n_all, min(n_all, 48), replace=False)
    at = int(rng.integers(0, 3)); gm = int(rng.integers(0, 2))
    use_prof = gm == 0 and rng.random() < 0.4
    kw = dict(align_type=at, gap_model=gm, match=int(rng.integers(-30, 3)), mismatch=int(rng.integers(0, 40)), n_score=int(rng.integers(-5, 8)),
              ins=int(rng.integers(0, 40)), dele=int(rng.integers(0, 40)), indel=int(rng.integers(0, 12)), hp_rule=int(rng.integers(0, 2)),
              tie_break=int(rng.integers(0, 2)), max_drops=int(rng.choice([0, 1, 3, 10, 40])),
              hp_ins_open=int(rng.integers(0, 20)), hp_ins_extend=int(rng.integers(0, 12)), ins_open=int(rng.integers(0, 20)), ins_extend=int(rng.integers(0, 12)))
    if rng.random() < 0.3:
        kw.update(stats_affine=1, stats_affine_open=int(rng.integers(0, 12)), stats_affine_extend=int(rng.integers(0, 6)))
    P = api.make_kband_params(**kw)
    pairs, ks, rev = [], [], []
    for p in pick:
        qo, ql = int(batch["q_off"][p]), int(batch["q_len"][p])
        to, tl = int(batch["t_off"][p]), int(batch["t_len"][p])
        if at == _lib.KB_EXTEND:
            ql, tl = min(ql, 1500), min(tl, 1500)
        pairs.append((np.ascontiguousarray(batch["q_arena"][qo:qo + ql]), np.ascontiguousarray(batch["t_arena"][to:to + tl])))
        base = abs(ql - tl) if rng.random() < 0.8 else 0
        ks.append(int(min(255, base + int(rng.choice([0, 1, 5, 10, 30, 80])))))
        rev.append(int(rng.random() < 0.4))
    from kband_cases import pack, random_profile
    prof = random_profile(rng, pairs) if use_prof else None
    b = pack(pairs, ks, profile=prof, reverse=np.array(rev, np.uint8))
    got, blocks, rc = ctx.kband_host(b, P)
    if rc != 0:
        return (0, "kband_rc", 0, rc), kw

    def one(i):
        q, t = pairs[i]
        pr = None
        if prof is not None:
            o = int(b["q_off"][i])
            pr = prof[o * 11:(o + len(q)) * 11]
        return oracle.kband_align(q, t, ks[i], P, profile=pr, reverse=bool(rev[i]))
    want = list(pool.map(one, range(len(pairs))))
    for i, (w, wb) in enumerate(want):
        g = got[i]
        if int(g["status"]) == _lib.PAIR_TOO_LONG:
            continue
        if int(g["status"]) != w["status"]:
            return (int(pick[i]), "kband_status", w["status"], int(g["status"]), ks[i], rev[i]), kw
        if w["status"] != 0:
            continue
        for fld in FIELDS:
            if int(g[fld]) != int(w[fld]):
                return (int(pick[i]), "kband_" + fld, int(w[fld]), int(g[fld]), ks[i], rev[i]), kw
        kb = blocks[int(g["blk_off"]):int(g["blk_off"]) + int(g["n_blk"])]
        if len(kb) != len(wb) or not (kb == wb).all():
            return (int(pick[i]), "kband_blocks", len(wb), len(kb), ks[i], rev[i]), kw
    return None, kw


def check_qv(rng, api, oracle, _lib, ctx, pool, batch, kw, local=False):
    """The round's batch again with the costs of a quality-value scorer (gba_batch.q_profile, random values, some outside
    +-127, or +-63 with affine gaps) against gbo_align_qv: linear and affine gaps, the k-band geometry; local: under the
    Local end rule (the CCS call site's combination)."""
    if local:
        kw = dict(kw, align_type=_lib.ALIGN_LOCAL)
        ctx.set_scoring(api.make_scoring(**kw))
    n = int(batch["n_pairs"])
    na = len(batch["q_arena"])
    prof = rng.integers(-20, 60, (na, 11)).astype(np.int16)
    wild = rng.random(na) < 0.01
    prof[wild, int(rng.integers(0, 11))] = int(rng.choice([-400, 500]))
    prof = prof.reshape(-1)
    pairs, blocks, rc = ctx.refine_host(dict(batch, q_profile=prof))
    if rc != 0:
        return (0, "qv_rc", 0, rc)
    P = oracle.make_params(**kw)

    def one(p):
        qo, ql = int(batch["q_off"][p]), int(batch["q_len"][p])
        to, tl = int(batch["t_off"][p]), int(batch["t_len"][p])
        bo, nb = int(batch["blk_off"][p]), int(batch["n_blk"][p])
        if qo + ql > na or to + tl > len(batch["t_arena"]):
            return None
        return oracle.align_qv(batch["q_arena"][qo:qo + ql], batch["t_arena"][to:to + tl], batch["blocks"][bo:bo + nb], P, prof[qo * 11:(qo + ql) * 11])
    want = list(pool.map(one, range(n)))
    for p in range(n):
        if want[p] is None:
            continue
        w, wb = want[p]
        g = pairs[p]
        if int(g["status"]) in (_lib.PAIR_TOO_WIDE, _lib.PAIR_TOO_LONG):
            continue
        if int(g["status"]) != w["status"]:
            return (p, "qv_status", w["status"], int(g["status"]))
        if w["status"] != 0:
            continue
        for fld in FIELDS:
            if int(g[fld]) != int(w[fld]):
                return (p, "qv_" + fld, int(w[fld]), int(g[fld]))
        kb = blocks[int(g["blk_off"]):int(g["blk_off"]) + int(g["n_blk"])]
        if len(kb) != len(wb) or not (kb == wb).all():
            return (p, "qv_blocks", len(wb), len(kb))
    return None


def compare_with_oracle(_lib, oracle, pool, batch, P, pairs, blocks, tag=""):
    """Every pair result and block of a host run against the oracle's; None or the first difference."""
    n = int(batch["n_pairs"])
    chunk = max(1, n // 16)
    futs = [pool.submit(oracle.align_batch, batch, P, f, min(chunk, n - f)) for f in range(0, n, chunk)]
    ores, oblk = [], []
    for f in futs:
        r, b = f.result()
        ores += list(r); oblk += b
    for p in range(n):
        o, k = ores[p], pairs[p]
        ko = int(k["status"])
        if ko in (_lib.PAIR_TOO_WIDE, _lib.PAIR_TOO_LONG):
            continue
        if int(o["status"]) != ko:
            return (p, tag + "status", int(o["status"]), ko)
        if ko != 0:
            continue
        for fld in FIELDS:
            if int(o[fld]) != int(k[fld]):
                return (p, tag + fld, int(o[fld]), int(k[fld]))
        kb = blocks[int(k["blk_off"]):int(k["blk_off"]) + int(k["n_blk"])]
        ob = oblk[p]
        if len(kb) != len(ob) or not (np.array_equal(kb["q_pos"], ob["q_pos"]) and np.array_equal(kb["t_pos"], ob["t_pos"])
                                      and np.array_equal(kb["length"], ob["length"])):
            return (p, tag + "blocks", len(ob), len(kb))
    return None


def check_local(api, oracle, _lib, ctx, pool, batch, kw):
    """The round's batch under the Local end rule (free start, end in the cheapest cell of the band) against the oracle's."""
    ctx.set_scoring(api.make_scoring(**kw, align_type=_lib.ALIGN_LOCAL))
    pairs, blocks, rc = ctx.refine_host(batch)
    if rc != 0:
        return (0, "local_rc", 0, rc)
    return compare_with_oracle(_lib, oracle, pool, batch, oracle.make_params(**kw, align_type=oracle.ALIGN_LOCAL), pairs, blocks, "local_")


def check_packed(rng, api, _lib, ctx, batch, want, want_blocks):
    """The round's batch with its target arena resident as the reference (packed 2 bits per base plus a mask bit, whatever
    letters it holds) and the targets given as windows of it, the reads 2 bits per base when they have no other letters:
    the same bytes as from letters and explicit targets; then the same comparison with random strand flags."""
    n = int(batch["n_pairs"])
    try:
        packed = api.pack_reads_2bit(batch)
    except ValueError:
        packed = batch
    ctx.set_reference(batch["t_arena"])
    try:
        win = {k: v for k, v in packed.items() if k != "t_arena"}
        got, got_blocks, rc = ctx.refine_host(win)
        if rc != 0 or got.tobytes() != want.tobytes() or got_blocks.tobytes() != want_blocks.tobytes():
            diff = [p for p in range(n) if got[p].tobytes() != want[p].tobytes()]
            return (diff[0] if diff else 0, "packed windows", rc, len(diff))
        flags = rng.integers(0, 2, n).astype(np.uint8)
        a, ab, rca = ctx.refine_host(dict(batch, t_rc=flags))
        b, bb, rcb = ctx.refine_host(dict(win, t_rc=flags))
        if rca != rcb or a.tobytes() != b.tobytes() or ab.tobytes() != bb.tobytes():
            diff = [p for p in range(n) if a[p].tobytes() != b[p].tobytes()]
            return (diff[0] if diff else 0, "packed windows, both strands", rcb, len(diff))
    finally:
        ctx.set_reference(None)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import util
    from blasr_amd import api, synth, _lib, emit
    from oracle import oracle
    rng = np.random.default_rng(args.seed)
    # three contexts used in turn: the library's defaults, one whose narrow affine wave class always runs four pairs per
    # wave (by default only batches of 10 240 pairs or more do, which these rounds never reach)
    os.environ["GBA_QUAD_AFFINE_MIN"] = "0"
    ctx_quad = api.Context(0)
    del os.environ["GBA_QUAD_AFFINE_MIN"]
    # and one that runs every batch on the lane-per-pair kernels as far as its pairs fit their classes (by default only
    # batches of 20 480 pairs or more do, affine 57 344)
    os.environ["GBA_WAVE_PAIRS"] = os.environ["GBA_WAVE_PAIRS_AFFINE"] = "1"
    ctx_lanes = api.Context(0)
    del os.environ["GBA_WAVE_PAIRS"], os.environ["GBA_WAVE_PAIRS_AFFINE"]
    ctxs = (api.Context(0), ctx_quad, ctx_lanes)
    pool = ThreadPoolExecutor(16)
    t_end = time.time() + args.seconds
    rounds = pairs_done = fails = unsupported = cigar_rounds = sdp_rounds = kband_rounds = qv_rounds = packed_rounds = local_rounds = 0
    hist = {}
    modes = {}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    while time.time() < t_end:
        ctx = ctxs[(rounds >> 1) % 3]
        kw = random_scoring(rng, _lib)
        batch = random_adversarial(rng, util) if rng.random() < 0.35 else random_synth(rng, synth)
        n = int(batch["n_pairs"])
        try:
            ctx.set_scoring(api.make_scoring(**kw))
        except api.GbaError as e:
            if e.code == _lib.E_UNSUPPORTED:
                unsupported += 1
                continue
            raise
        cig = None
        if rounds % 2 == 1:
            cig_flags = int(rng.integers(0, 4))
            pairs, blocks, ops, off, rc = ctx.refine_host_cigar(batch, flags=cig_flags)
            cig = (ops, off, cig_flags)
        else:
            pairs, blocks, rc = ctx.refine_host(batch)
        assert rc == 0
        P = oracle.make_params(**kw)
        chunk = max(1, n // 16)
        futs = [pool.submit(oracle.align_batch, batch, P, f, min(chunk, n - f)) for f in range(0, n, chunk)]
        ores, oblk = [], []
        for f in futs:
            r, b = f.result()
            ores += list(r); oblk += b
        bad = None
        for p in range(n):
            o, k = ores[p], pairs[p]
            ko = int(k["status"])
            if ko in (_lib.PAIR_TOO_WIDE, _lib.PAIR_TOO_LONG):   # kernel limits the oracle does not have
                continue
            if int(o["status"]) != ko:
                bad = (p, "status", int(o["status"]), ko); break
            if ko != 0:
                continue
            for fld in FIELDS:
                if int(o[fld]) != int(k[fld]):
                    bad = (p, fld, int(o[fld]), int(k[fld])); break
            if bad:
                break
            kb = blocks[int(k["blk_off"]):int(k["blk_off"]) + int(k["n_blk"])]
            ob = oblk[p]
            if len(kb) != len(ob) or not (np.array_equal(kb["q_pos"], ob["q_pos"]) and np.array_equal(kb["t_pos"], ob["t_pos"])
                                          and np.array_equal(kb["length"], ob["length"])):
                bad = (p, "blocks", len(ob), len(kb)); break
            qp, tp = int(k["q_pos"]), int(k["t_pos"])
            want_iv = (qp, qp, tp, tp) if len(ob) == 0 else (
                qp + int(ob[0]["q_pos"]), qp + int(ob[-1]["q_pos"]) + int(ob[-1]["length"]),
                tp + int(ob[0]["t_pos"]), tp + int(ob[-1]["t_pos"]) + int(ob[-1]["length"]))
            got_iv = tuple(int(k[f]) for f in ("q_start", "q_end", "t_start", "t_end"))
            if got_iv != want_iv:
                bad = (p, "interval", want_iv, got_iv); break
        if not bad and rounds % 4 == 0:
            # the same batch without a block arena: the same pair results, nothing refused for space
            only, rc_iv = ctx.refine_host_intervals(batch)
            if rc_iv != 0 or ctx.blocks_needed != len(blocks) or only.tobytes() != pairs.tobytes():
                diff = [p for p in range(n) if only[p].tobytes() != pairs[p].tobytes()]
                bad = (diff[0] if diff else 0, "no-arena run", rc_iv, len(diff))
        if not bad and cig is not None:
            bad = check_cigars(rng, emit, _lib, batch, pairs, blocks, cig[0], cig[1], cig[2])
            cigar_rounds += 1
        if not bad and rounds % 3 == 2:
            bad, skw = check_sdp(rng, api, oracle, _lib, ctx, pool, batch)
            sdp_rounds += 1
            if bad:
                kw = dict(kw, sdp=skw)
        if not bad and rounds % 4 == 3:
            bad = check_qv(rng, api, oracle, _lib, ctx, pool, batch, kw)
            qv_rounds += 1
        if not bad and rounds % 3 == 1:
            bad, kkw = check_kband(rng, api, oracle, _lib, ctx, pool, batch)
            kband_rounds += 1
            if bad:
                kw = dict(kw, kband=kkw)
        if not bad and rounds % 5 == 0 and cig is None:
            bad = check_packed(rng, api, _lib, ctx, batch, pairs, blocks)
            packed_rounds += 1
        if not bad and rounds % 5 == 1 and kw["mode"] == _lib.MODE_LINEAR:
            bad = check_local(api, oracle, _lib, ctx, pool, batch, kw)
            if not bad and rounds % 10 == 1:
                bad = check_qv(rng, api, oracle, _lib, ctx, pool, batch, kw, local=True)
            local_rounds += 1
        rounds += 1; pairs_done += n
        for v, c in zip(*np.unique(pairs["status"], return_counts=True)):
            hist[int(v)] = hist.get(int(v), 0) + int(c)
        modes[kw["mode"]] = modes.get(kw["mode"], 0) + n
        if bad:
            fails += 1
            p = bad[0]
            qo, ql = int(batch["q_off"][p]), int(batch["q_len"][p])
            to, tl = int(batch["t_off"][p]), int(batch["t_len"][p])
            bo, nb = int(batch["blk_off"][p]), int(batch["n_blk"][p])
            rep = {"scoring": kw, "what": [str(x) for x in bad],
                   "q": bytes(batch["q_arena"][qo:qo + ql]).decode("latin1"),
                   "t": bytes(batch["t_arena"][to:to + tl]).decode("latin1"),
                   "blocks": [[int(b["q_pos"]), int(b["t_pos"]), int(b["length"])] for b in batch["blocks"][bo:bo + nb]]}
            with open(os.path.join(ROOT, "gpurun_out", "fuzz_fail_%d.json" % fails), "w") as f:
                json.dump(rep, f)
            print("MISMATCH", bad, kw, flush=True)
            if fails >= 5:
                break
        if rounds % 20 == 0:
            print("rounds %d pairs %d fails %d unsupported-scoring %d" % (rounds, pairs_done, fails, unsupported), flush=True)
    print("DONE rounds %d pairs %d fails %d unsupported-scoring %d" % (rounds, pairs_done, fails, unsupported), flush=True)
    print("status histogram", hist, "pairs per mode", modes, "rounds with CIGAR check", cigar_rounds, "rounds with guide production", sdp_rounds, "rounds with fixed-band aligners", kband_rounds, "rounds with quality-value costs", qv_rounds,
          "rounds with packed reads and reference windows", packed_rounds, "rounds under the Local end rule", local_rounds, flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
