"""Synthetic cases for the fixed-band aligners (shared by the CPU and GPU tests)."""
import numpy as np

FIELDS = ("status", "dp_score", "score", "q_pos", "t_pos", "n_blk", "n_match", "n_mismatch",
          "n_ins", "n_del", "rows", "cols", "ncells")
ACGT = np.frombuffer(b"ACGT", np.uint8)


def _target(rng, n, homopolymers):
    if not homopolymers:
        t = rng.choice(ACGT, n)
    else:       # runs of one to six equal bases
        t = np.repeat(rng.choice(ACGT, n), rng.integers(1, 7, n))[:n]
    if n > 20 and rng.random() < 0.3:
        t[rng.integers(0, n, 3)] = ord("N")
    return t.astype(np.uint8)


def _read(rng, t, rate, homopolymers):
    out = []
    for c in t:
        r = rng.random()
        if r < rate * 0.5:          # insertion (in homopolymer mode mostly a repeat of the base)
            out.append(c if homopolymers and rng.random() < 0.7 else rng.choice(ACGT)); out.append(c)
        elif r < rate * 0.8:        # deletion
            continue
        elif r < rate:              # substitution
            out.append(rng.choice(ACGT))
        else:
            out.append(c)
    q = np.array(out, dtype=np.uint8)
    if len(q) and rng.random() < 0.2:
        q[rng.integers(0, len(q))] = ord("a")       # lower case maps like upper case
    return q


def make_pairs(rng, n, length, flank=False, homopolymers=False):
    """n pairs (query, target) as uint8 arrays and a band per pair that reaches the end cell."""
    pairs, ks = [], []
    for _ in range(n):
        t = _target(rng, int(rng.integers(length[0], length[1])), homopolymers)
        q = _read(rng, t, float(rng.choice([0.02, 0.12, 0.25])), homopolymers)
        if flank:       # the target runs on at both ends
            t = np.concatenate([_target(rng, int(rng.integers(0, 12)), False), t, _target(rng, int(rng.integers(0, 12)), False)])
        pairs.append((q, t))
        ks.append(abs(len(q) - len(t)) + int(rng.integers(0, 30)))
    return pairs, ks


def random_profile(rng, pairs):
    """11 costs per query base in the order the pairs are packed: substitution against A, C, G, T, N,
    insertion, deletion of a target A, C, G, T, N behind the base."""
    prof = []
    for q, _ in pairs:
        p = np.zeros((len(q), 11), np.int16)
        p[:, :5] = rng.integers(1, 25, (len(q), 5))
        code = np.array([{65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}.get(int(c), 4) for c in q], dtype=np.int64)
        p[np.arange(len(q)), code] = rng.integers(-6, 1, len(q))
        p[:, 5] = rng.integers(1, 20, len(q))
        p[:, 6:] = rng.integers(1, 20, (len(q), 5))
        prof.append(p.reshape(-1))
    return np.concatenate(prof) if prof else np.zeros(0, np.int16)


def pack(pairs, ks, profile=None, reverse=None):
    n = len(pairs)
    q_len = np.array([len(q) for q, _ in pairs], np.uint32)
    t_len = np.array([len(t) for _, t in pairs], np.uint32)
    q_off = np.zeros(n, np.uint64); t_off = np.zeros(n, np.uint64)
    q_off[1:] = np.cumsum(q_len[:-1].astype(np.uint64)); t_off[1:] = np.cumsum(t_len[:-1].astype(np.uint64))
    b = {"n_pairs": n,
         "q_arena": np.concatenate([q for q, _ in pairs] + [np.zeros(0, np.uint8)]).astype(np.uint8),
         "t_arena": np.concatenate([t for _, t in pairs] + [np.zeros(0, np.uint8)]).astype(np.uint8),
         "q_off": q_off, "t_off": t_off, "q_len": q_len, "t_len": t_len, "k": np.array(ks, np.int32)}
    if profile is not None:
        b["profile"] = np.ascontiguousarray(profile, np.int16)
    if reverse is not None:
        b["reverse"] = np.ascontiguousarray(reverse, np.uint8)
    return b
