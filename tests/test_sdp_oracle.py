"""The guide producer's CPU restatement (oracle/sdp_oracle.c) against a direct, quadratic Python
statement of its definition on small cases, and its properties on synthetic pairs.  Parity with
libcpp's SDPAlign is unpinned (the reference tree holds neither its source nor any vector)."""
import numpy as np
import pytest


def _code(c):
    return {65: 0, 67: 1, 71: 2, 84: 3}.get(c & 0xDF, 4)


def _brute(q, t, k, ins, dele, match, max_fragments):
    """Maximal exact matches by scanning every diagonal, then the chain by its definition."""
    Q, T = len(q), len(t)
    frags = []
    for d in range(-(Q - 1), T):
        x = max(0, -d); y = x + d
        run = 0
        while x <= Q and y <= T:
            same = x < Q and y < T and _code(q[x]) < 4 and _code(q[x]) == _code(t[y])
            if same:
                run += 1
            else:
                if run >= k:
                    frags.append((x - run, y - run, run))
                run = 0
            x += 1; y += 1
    frags.sort()
    hits = sum(l - k + 1 for _, _, l in frags)
    if len(frags) > max_fragments or hits > 64 * max_fragments:
        return 8, [], 0
    if not frags:
        return 1, [], 0
    score, prev = [], []
    for f, (x, y, l) in enumerate(frags):
        best, bp = 0, -1
        for p in range(f):
            px, py, pl = frags[p]
            o = max(px + pl - x, py + pl - y, 0)
            if o >= pl:
                continue
            dd = (y - x) - (py - px)
            g = score[p] - (dd * dele if dd >= 0 else -dd * ins) - o * match
            if g > best:
                best, bp = g, p
        score.append(l * match + best); prev.append(bp)
    end = max(range(len(frags)), key=lambda f: (score[f], -f))
    chain = []
    f = end
    while f >= 0:
        chain.append(f); f = prev[f]
    chain.reverse()
    out = []
    for i, f in enumerate(chain):
        x, y, l = frags[f]
        if i + 1 < len(chain):
            nx, ny, _ = frags[chain[i + 1]]
            l -= max(x + l - nx, y + l - ny, 0)
        out.append((x, y, l))
    return 0, out, score[end]


def _mutate(rng, s, rate):
    out = []
    for c in s:
        r = rng.random()
        if r < rate / 3:
            continue
        if r < 2 * rate / 3:
            out.append(rng.choice(list(b"ACGT")))
        if r < rate:
            out.append(rng.choice(list(b"ACGT")))
        else:
            out.append(c)
    return bytes(out)


@pytest.mark.parametrize("k", [4, 6, 11])
def test_against_the_definition(oracle, k):
    rng = np.random.default_rng(100 + k)
    for case in range(60):
        n = int(rng.integers(0, 260))
        alphabet = list(b"ACGT") if case % 3 else list(b"ACGTNacgtR")
        t = bytes(rng.choice(alphabet, n).tolist())
        q = _mutate(rng, t, 0.12) if case % 4 else bytes(rng.choice(alphabet, int(rng.integers(0, 200))).tolist())
        if case % 7 == 0:
            q = q[:40] + q[:40] + q      # a repeat: fragments that overlap in the target
        maxf = 64 if case % 5 else 8
        P = oracle.make_sdp_params(k=k, ins=5, dele=10, match=5, max_fragments=maxf)
        st, blks, score, nfrag = oracle.sdp_chain(q, t, P)
        est, eb, escore = _brute(q, t, k, 5, 10, 5, maxf)
        assert st == est, case
        if st == 0:
            assert [tuple(int(v) for v in b) for b in blks] == eb, case
            assert score == escore


def test_chain_properties_on_synthetic_reads(oracle, synth):
    batch = synth.generate(12, read_len=4000, seed=8, n_frac=0.01)
    P = oracle.make_sdp_params()
    st, nb, score, nfrag, blks = oracle.sdp_chain_batch(batch, P)
    assert (st == 0).all()
    for p in range(12):
        q, t, guide = synth.pair(batch, p)
        b = blks[p]
        assert len(b) == nb[p] > 40
        # gap-free exact matches, strictly increasing, never overlapping
        for i in range(len(b)):
            x, y, l = int(b[i]["q_pos"]), int(b[i]["t_pos"]), int(b[i]["length"])
            assert l >= 1 and bytes(q[x:x + l]).upper() == bytes(t[y:y + l]).upper()
            if i:
                px, py, pl = (int(b[i - 1][f]) for f in ("q_pos", "t_pos", "length"))
                assert px + pl <= x and py + pl <= y
        # the chain follows the diagonal the read was drawn from: it covers most of what the truth-derived guide covers
        covered = sum(int(v) for v in b["length"])
        truth = sum(int(g[2]) for g in guide if int(g[2]) >= 11)
        assert covered >= 0.9 * truth


def test_statuses(oracle):
    P = oracle.make_sdp_params(k=11, max_fragments=16)
    assert oracle.sdp_chain(b"ACGT", b"ACGTACGTACGTACGT", P)[0] == oracle.SDP_EMPTY          # shorter than a tuple
    assert oracle.sdp_chain(b"A" * 30 + b"C" * 30, b"G" * 40 + b"T" * 40, P)[0] == oracle.SDP_EMPTY
    assert oracle.sdp_chain(b"A" * 300, b"A" * 300, P)[0] == oracle.SDP_TOO_MANY
    assert oracle.sdp_chain(b"N" * 50, b"N" * 50, P)[0] == oracle.SDP_EMPTY
    st, b, score, _ = oracle.sdp_chain(b"ACGTTGCAAGGCTTAGCATCGA", b"ACGTTGCAAGGCTTAGCATCGA", P)
    assert st == 0 and [tuple(int(v) for v in x) for x in b] == [(0, 0, 22)] and score == 110
