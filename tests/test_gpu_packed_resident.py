"""The packed forms in device memory: the resident reference (2 bits per base and a mask bit for every letter that is
not one of ACGT / acgt) and reads that stay 2 bits per base (gba_batch.q_packed), through every entry point that takes a
batch.  The checks are against the same calls on plain letters, which the other tests pin to the oracles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_COMP = np.full(256, ord("N"), np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCATGCA"):
    _COMP[_a] = _b
_UP = np.full(256, ord("N"), np.uint8)
for _a, _b in zip(b"ACGTacgt", b"ACGTACGT"):
    _UP[_a] = _b


def _spell(ref, start, length, rc):
    w = ref[start:start + length]
    return _COMP[w][::-1] if rc else _UP[w]


def _odd_letters(ref, rng):
    """Lower-case stretches, IUPAC codes and single n's sprinkled over a copy of the reference."""
    ref = ref.copy()
    for s in rng.integers(0, len(ref) - 5000, 40):
        ref[s:s + 3000] |= 0x20                                        # soft-masked repeats
    at = rng.integers(0, len(ref), len(ref) // 300)
    ref[at] = rng.choice(np.frombuffer(b"RYKMSWBDHVnNx-", np.uint8), len(at))
    return ref


def test_reference_with_other_letters_as_windows_equals_spelled_out_targets(synth):
    from blasr_amd import api
    rng = np.random.default_rng(77)
    ref = _odd_letters(synth.reference(300_000, 9, n_frac=0.01), rng)
    batch = synth.compact_blocks(synth.generate(600, read_len=900, len_jitter=500, seed=78, ref=ref, ref_len=len(ref)))
    n = 600
    assert 200 < int(batch["win_rc"].sum()) < 400
    # the explicit form of the same batch: every target spelled out by numpy, upper case, N for every other letter
    t_len = batch["t_len"].astype(np.int64)
    t_off = np.zeros(n, np.int64); t_off[1:] = np.cumsum(t_len[:-1])
    t_arena = np.concatenate([_spell(ref, int(batch["win_start"][p]), int(t_len[p]), int(batch["win_rc"][p])) for p in range(n)])
    assert (t_arena == ord("N")).sum() > 200
    explicit = dict(batch, t_arena=t_arena, t_off=t_off.astype(np.uint64))
    for k in ("win_start", "win_rc"):
        explicit.pop(k)
    ctx = api.Context(0)
    try:
        ctx.set_scoring(api.make_scoring())
        want, wblocks, rc = ctx.refine_host(explicit)
        assert rc == 0 and int((want["status"] == 0).sum()) >= n - 5
        ctx.set_reference(ref)
        got, gblocks, rc = ctx.refine_host(synth.as_windows(batch))
        assert rc == 0 and got.tobytes() == want.tobytes() and gblocks.tobytes() == wblocks.tobytes()
        # and the guide producer reads the same bases: chains from windows = chains from spelled-out targets
        sp = api.make_sdp_params()
        seq = lambda b: {k: v for k, v in b.items() if k not in ("blocks", "blk_off", "n_blk")}
        a = ctx.sdp_host(seq(explicit), sp)
        b = ctx.sdp_host(seq(synth.as_windows(batch)), sp)
        assert a[5] == 0 and b[5] == 0
        for x, y in zip(a[:5], b[:5]):
            assert x.tobytes() == y.tobytes()
        assert int((a[0] == 0).sum()) >= 400               # short reads at 15 % error: some have no chain
    finally:
        ctx.close()


def test_windows_at_both_ends_of_the_reference_on_both_strands(synth):
    """Windows that start within the first packed word of the reference and windows that end with its last base, forward
    and reverse, at every alignment of the start to the 16-base words; the reference length is not a multiple of 16 or 32."""
    from blasr_amd import api
    rng = np.random.default_rng(79)
    ref = _odd_letters(synth.reference(100_003, 11), rng)
    L = len(ref)
    wins = [(s, 400 + 7 * s, rc) for s in range(0, 35) for rc in (0, 1)]
    wins += [(L - (400 + 5 * e), 400 + 5 * e, rc) for e in range(0, 35) for rc in (0, 1)]
    wins += [(0, 17, 0), (0, 17, 1), (L - 16, 16, 1), (L - 33, 33, 0), (5, 1, 0), (L - 1, 1, 1)]
    n = len(wins)
    targets = [_spell(ref, s, l, rc) for s, l, rc in wins]
    # reads: each target with a few edits, so that chains exist and the band is used
    reads = []
    for t in targets:
        q = t.copy()
        q[q == ord("N")] = ord("A")
        if len(q) > 100:
            for at in sorted(rng.integers(20, len(q) - 20, 6), reverse=True):
                q = np.concatenate([q[:at], q[at + 1:]]) if at & 1 else np.concatenate([q[:at], np.frombuffer(b"G", np.uint8), q[at:]])
        reads.append(q)
    q_len = np.array([len(q) for q in reads], np.uint32)
    q_off = np.zeros(n, np.uint64); q_off[1:] = np.cumsum(q_len[:-1])
    t_len = np.array([l for _, l, _ in wins], np.uint32)
    t_off = np.zeros(n, np.uint64); t_off[1:] = np.cumsum(t_len[:-1])
    common = dict(n_pairs=n, q_arena=np.concatenate(reads), q_off=q_off, q_len=q_len, t_len=t_len)
    explicit = dict(common, t_arena=np.concatenate(targets), t_off=t_off)
    windows = dict(common, t_off=np.array([s for s, _, _ in wins], np.uint64), t_rc=np.array([rc for _, _, rc in wins], np.uint8))
    ctx = api.Context(0)
    try:
        ctx.set_scoring(api.make_scoring())
        ctx.set_reference(ref)
        sp = api.make_sdp_params()
        a = ctx.sdp_host(explicit, sp)
        b = ctx.sdp_host(windows, sp)
        for x, y in zip(a[:5], b[:5]):
            assert x.tobytes() == y.tobytes()
        assert int((a[0] == 0).sum()) >= 140
        ctx.set_guide_producer(sp)
        want, wblocks, rc = ctx.refine_host(explicit)
        got, gblocks, rc2 = ctx.refine_host(windows)
        assert rc == 0 and rc2 == 0 and got.tobytes() == want.tobytes() and gblocks.tobytes() == wblocks.tobytes()
        assert int((want["status"] == 0).sum()) >= 140
        # a window that reaches past the end of the reference is refused for its pair, as before
        over = dict(windows, t_off=windows["t_off"].copy())
        over["t_off"][3] = L - 10
        got, _, rc = ctx.refine_host(over)
        assert rc == 0 and int(got["status"][3]) != 0 and int(got["status"][4]) == int(want["status"][4])
    finally:
        ctx.close()


def test_packed_reads_resident_on_the_device(synth):
    """gba_submit_device, gba_submit_device_noalloc and gba_sdp_device on a batch whose reads are 2 bits per base in
    device memory, against the same calls on letters; the host SDP entry point too."""
    import torch
    from blasr_amd import _lib, api
    host = synth.compact_blocks(synth.generate(500, read_len=1800, len_jitter=900, seed=93, decoy_every=9))
    packed = api.pack_reads_2bit(host)
    n = 500
    ctx = api.Context(0)
    try:
        def run(db, **kw):
            pairs = torch.zeros(n * _lib.PAIR_RESULT_DT.itemsize, dtype=torch.uint8, device="cuda:0")
            blocks = torch.zeros((int(host["q_len"].astype(np.int64).sum()) + n, 3), dtype=torch.int32, device="cuda:0")
            rc, needed = ctx.refine_device(db, pairs, blocks, **kw)
            assert rc == 0
            return pairs.cpu().numpy().tobytes(), blocks.cpu().numpy()[:needed].tobytes()
        dl, dp = api.to_device(host), api.to_device(packed)
        assert dp["q_arena"].numel() * 4 < dl["q_arena"].numel() + 64 and dp["q_packed"] == 1
        for kw in (dict(), dict(mode=_lib.MODE_AFFINE, band=16), dict(band=120)):
            ctx.set_scoring(api.make_scoring(**kw))
            want = run(dl)
            assert run(dp) == want
            assert run(dp, no_alloc=True) == want
        # a packed arena that does not start on a 4-byte boundary is refused
        shifted = dict(dp, q_arena=torch.cat([torch.zeros(1, dtype=torch.uint8, device="cuda:0"), dp["q_arena"]])[1:])
        if shifted["q_arena"].data_ptr() % 4:
            with pytest.raises(_lib.GbaError):
                run(shifted)
        # guides from the packed reads, on the device and from host buffers
        sp = api.make_sdp_params()
        seq = lambda b: {k: v for k, v in b.items() if k not in ("blocks", "blk_off", "n_blk")}
        gl, gp = ctx.sdp_device(seq(dl), sp), ctx.sdp_device(seq(dp), sp)
        assert gl["rc"] == 0 and gp["rc"] == 0 and gl["blocks_needed"] == gp["blocks_needed"]
        for k in ("status", "chain_score", "n_blk", "blk_off"):
            assert torch.equal(gl[k], gp[k]), k
        assert torch.equal(gl["blocks"][:gl["blocks_needed"]], gp["blocks"][:gp["blocks_needed"]])
        hl, hp = ctx.sdp_host(seq(host), sp), ctx.sdp_host(seq(packed), sp)
        for x, y in zip(hl[:5], hp[:5]):
            assert x.tobytes() == y.tobytes()
        assert np.array_equal(hl[0], gl["status"].cpu().numpy()) and int((hl[0] == 0).sum()) > 40
        # sequences in, refined alignments out, all on the device
        ctx.set_scoring(api.make_scoring())
        ctx.set_guide_producer(sp)
        assert run(seq(dp)) == run(seq(dl))
    finally:
        ctx.close()


def test_resident_reference_footprint(synth):
    """3 bits per base (2 for the code, 1 for the mask) plus a few words of slack: hipMemGetInfo before and after."""
    import torch
    from blasr_amd import api
    ctx = api.Context(0)
    try:
        ref = synth.reference(synth.CHR20_LEN, n_frac=0.01)
        ctx.set_reference(synth.reference(1000, 3))          # the staging chunk and the kernels are loaded now
        ctx.set_reference(None)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        ctx.set_reference(ref)
        torch.cuda.synchronize()
        used = free0 - torch.cuda.mem_get_info(0)[0]
        assert 0 < used <= len(ref) * 3 // 8 + (16 << 20), used       # the allocator's granularity on top; letters would take len(ref)
    finally:
        ctx.close()
