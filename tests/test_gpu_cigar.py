"""CIGAR operations written by the device (gba_results.cigar_*), through the C-ABI, against the CPU
oracle's column-by-column restatement (oracle/cigar_oracle.c) applied to the same refined blocks and
sequences; the host writer blasr_amd.emit.cigar_ops must agree with both.  The reference's own writers
are not in its tree (parity unpinned): what is checked is that the three forms agree operation for
operation, and that the operations account for every base."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLAG_SETS = [0, 1, 2, 3]      # bit 0: = / X instead of M, bit 1: keep adjacent indels


@pytest.fixture(scope="module")
def api():
    from blasr_amd import api as a
    return a


def _decode(ops):
    from blasr_amd import _lib
    return [(int(w) >> 4, _lib.CIGAR_OPS[int(w) & 15]) for w in ops]


_CODE_LETTER = np.full(256, ord("N"), np.uint8)
for _c in b"ACGT":
    _CODE_LETTER[_c] = _c
    _CODE_LETTER[_c + 32] = _c


def _by_code(seq):
    """The sequence as the kernels see it: case folded, every letter outside ACGT an N."""
    return _CODE_LETTER[np.frombuffer(bytes(seq), dtype=np.uint8)].tobytes()


def _check(api, synth, batch, pairs, blocks, ops, off, flags, expect_all_ok=True):
    from blasr_amd import _lib, emit
    from oracle import oracle as cpu_oracle
    n = int(batch["n_pairs"])
    assert int(off[0]) == 0 and int(off[n]) == len(ops)
    assert (np.diff(off.astype(np.int64)) >= 0).all()
    n_ok = 0
    for p in range(n):
        got = _decode(ops[int(off[p]):int(off[p + 1])])
        k = pairs[p]
        if int(k["status"]) != 0:
            assert got == []
            continue
        n_ok += 1
        q, t, _ = synth.pair(batch, p)
        kb = blocks[int(k["blk_off"]):int(k["blk_off"]) + int(k["n_blk"])]
        fl = dict(use_seq_match=bool(flags & _lib.CIGAR_SEQ_MATCH), allow_adjacent_indels=bool(flags & _lib.CIGAR_ADJACENT_INDELS))
        want = cpu_oracle.cigar_ops(kb, q, t, int(k["q_pos"]), int(k["t_pos"]), **fl)
        assert got == want, (p, flags)
        assert emit.cigar_ops(kb, _by_code(q), _by_code(t), int(k["q_pos"]), int(k["t_pos"]), **fl) == want, (p, flags)
        # every base from the first block to the end of the last one is accounted for (gaps in front of the
        # first block and behind the last are not part of the block list), no two equal operations in a row
        assert sum(c for c, o in got if o in "M=XI") == int(kb[-1]["q_pos"]) + int(kb[-1]["length"]) - int(kb[0]["q_pos"])
        assert sum(c for c, o in got if o in "M=XD") == int(kb[-1]["t_pos"]) + int(kb[-1]["length"]) - int(kb[0]["t_pos"])
        assert all(a[1] != b[1] for a, b in zip(got, got[1:])) and all(c > 0 for c, _ in got)
        if flags == 3:       # unmerged, = / X: the same counts as the kernel's own statistics
            tot = {o: sum(c for c, oo in got if oo == o) for o in "=XID"}
            assert (tot["="], tot["X"]) == (int(k["n_match"]), int(k["n_mismatch"]))
            assert tot["I"] <= int(k["n_ins"]) and tot["D"] <= int(k["n_del"])
    if expect_all_ok:
        assert n_ok == n


@pytest.mark.parametrize("family", ["lane-per-pair", "wave-per-pair"])
def test_host_batches_all_flag_sets(api, synth, family, monkeypatch):
    v = "0" if family == "lane-per-pair" else "4000000000"
    monkeypatch.setenv("GBA_WAVE_PAIRS", v)
    monkeypatch.setenv("GBA_WAVE_PAIRS_AFFINE", v)
    ctx = api.Context(0)
    try:
        batch = synth.generate(150, read_len=3000, seed=77, len_jitter=1500, n_frac=0.01)
        from blasr_amd import _lib
        for kw in (dict(), dict(mode=_lib.MODE_AFFINE, band=16, affine_open=10, affine_extend=1)):
            ctx.set_scoring(api.make_scoring(**kw))
            for flags in FLAG_SETS:
                pairs, blocks, ops, off, rc = ctx.refine_host_cigar(batch, flags=flags)
                assert rc == 0 and ctx.cigar_needed == len(ops)
                _check(api, synth, batch, pairs, blocks, ops, off, flags, expect_all_ok=False)
    finally:
        ctx.close()


def test_parts_reverse_strand_windows_and_failed_pairs(api, synth, monkeypatch):
    """A batch run in several parts (operation offsets continue across parts), targets as windows of
    a resident reference on both strands, decoy candidates among them, and pairs that fail."""
    monkeypatch.setenv("GBA_PART_PAIRS", "256")
    ctx = api.Context(0)
    try:
        ref = synth.reference(2_000_000, n_frac=0.01)
        batch = synth.generate(1000, read_len=1200, seed=5, decoy_every=5, share_reads=True, ref=ref, len_jitter=600)
        batch["n_blk"] = batch["n_blk"].copy()
        batch["n_blk"][[3, 300, 999]] = 0                      # empty guides: no operations for these
        ctx.set_scoring(api.make_scoring())
        ctx.set_reference(ref)
        pairs, blocks, ops, off, rc = ctx.refine_host_cigar(synth.as_windows(batch), flags=1)
        assert rc == 0
        assert [int(pairs["status"][p]) for p in (3, 300, 999)] == [1, 1, 1]
        assert int(batch["win_rc"].sum()) > 100
        _check(api, synth, batch, pairs, blocks, ops, off, 1, expect_all_ok=False)
        assert int((pairs["status"] == 0).sum()) > 900
    finally:
        ctx.close()


def test_device_buffers_and_too_small_an_arena(api, synth):
    import torch
    from blasr_amd import _lib
    ctx = api.Context(0)
    try:
        batch = synth.generate(300, read_len=2000, seed=9)
        ctx.set_scoring(api.make_scoring())
        db = api.to_device(batch)
        n = 300
        out_pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
        out_blocks = torch.zeros((int(batch["q_len"].sum()) + n, 3), dtype=torch.int32, device="cuda:0")
        d_ops = torch.zeros(int(batch["q_len"].sum()) * 2, dtype=torch.int32, device="cuda:0")
        d_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda:0")
        for no_alloc in (False, True):
            d_ops.zero_(); d_off.zero_()
            rc, needed = ctx.refine_device(db, out_pairs, out_blocks, no_alloc=no_alloc, cigar=(d_ops, d_off, 3))
            torch.cuda.synchronize()
            assert rc == 0
            pairs = np.frombuffer(out_pairs.cpu().numpy().tobytes(), dtype=_lib.PAIR_RESULT_DT)
            blocks = np.frombuffer(out_blocks.cpu().numpy()[:needed].tobytes(), dtype=_lib.BLOCK_DT)
            ops = d_ops.cpu().numpy().view(np.uint32)[:ctx.cigar_needed]
            off = d_off.cpu().numpy().view(np.uint64)
            _check(api, synth, batch, pairs, blocks, ops, off, 3)
        full = ops.copy()
        # an arena that holds about half: NOSPACE, the count is still reported, pairs that fit are written
        half = int(off[150])
        small = torch.zeros(half, dtype=torch.int32, device="cuda:0")
        rc, _ = ctx.refine_device(db, out_pairs, out_blocks, cigar=(small, d_off, 3))
        torch.cuda.synchronize()
        assert rc == _lib.E_NOSPACE and ctx.cigar_needed == len(full)
        assert (small.cpu().numpy().view(np.uint32) == full[:half]).all()
        assert (d_off.cpu().numpy().view(np.uint64) == off).all()
        # no CIGAR wanted: nothing of it is touched
        rc, _ = ctx.refine_device(db, out_pairs, out_blocks)
        assert rc == 0 and ctx.cigar_needed == 0
    finally:
        ctx.close()


def test_lower_case_and_ambiguity_letters(api, synth):
    """Bases are compared by code: case does not matter, letters outside ACGT are all N."""
    ctx = api.Context(0)
    try:
        batch = synth.generate(64, read_len=1500, seed=21)
        rng = np.random.default_rng(3)
        for key in ("q_arena", "t_arena"):
            a = batch[key].copy()
            lower = rng.random(a.size) < 0.3
            a[lower] |= 32
            odd = rng.random(a.size) < 0.01
            a[odd] = rng.choice(np.frombuffer(b"RYKMrykmnN-", dtype=np.uint8), int(odd.sum()))
            batch[key] = a
        ctx.set_scoring(api.make_scoring())
        for flags in (1, 3):
            pairs, blocks, ops, off, rc = ctx.refine_host_cigar(batch, flags=flags)
            assert rc == 0
            _check(api, synth, batch, pairs, blocks, ops, off, flags, expect_all_ok=False)
        assert int((pairs["status"] == 0).sum()) > 50
    finally:
        ctx.close()


def test_empty_batch(api, synth):
    ctx = api.Context(0)
    try:
        ctx.set_scoring(api.make_scoring())
        batch = synth.generate(4, read_len=200, seed=1)
        empty = dict(batch, n_pairs=0)
        pairs, blocks, ops, off, rc = ctx.refine_host_cigar(empty, cigar_capacity=8)
        assert rc == 0 and len(ops) == 0 and int(off[0]) == 0
    finally:
        ctx.close()
