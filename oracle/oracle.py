"""ctypes wrapper over oracle/liboracle.so (the C restatement in gba_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
Parity with the reference binary is unpinned (see gba_oracle.c header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
# GBA_ORACLE_SO: another build of the same sources (the sanitizer build `make -C oracle asan`, tests/test_oracle_asan.py)
_ALT = os.environ.get("GBA_ORACLE_SO")

MODE_LINEAR, MODE_AFFINE, MODE_KBAND = 0, 1, 2
OK, EMPTY_GUIDE, BAD_GUIDE, UNREACHABLE, NOSPACE, END_OUT_OF_BAND = range(6)
NO_ALIGNMENT = 9
ALIGN_GLOBAL, ALIGN_LOCAL = 0, 1

BLOCK_DT = np.dtype([("q_pos", "<u4"), ("t_pos", "<u4"), ("length", "<u4")])


class Params(C.Structure):
    _fields_ = [
        ("sub", (C.c_int32 * 5) * 5),
        ("ins", C.c_int32), ("dele", C.c_int32),
        ("affine_open", C.c_int32), ("affine_extend", C.c_int32),
        ("stats_ins", C.c_int32), ("stats_del", C.c_int32),
        ("stats_affine_open", C.c_int32), ("stats_affine_extend", C.c_int32),
        ("band", C.c_int32), ("max_band", C.c_int32),
        ("mode", C.c_int32), ("tie_break", C.c_int32),
        ("kband_k", C.c_int32), ("stats_affine", C.c_int32), ("align_type", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("dp_score", C.c_int32), ("score", C.c_int32),
        ("q_pos", C.c_uint32), ("t_pos", C.c_uint32), ("n_blk", C.c_uint32),
        ("n_match", C.c_uint32), ("n_mismatch", C.c_uint32),
        ("n_ins", C.c_uint32), ("n_del", C.c_uint32),
        ("rows", C.c_uint32), ("cols", C.c_uint32),
        ("ncells", C.c_int64),
    ]


RESULT_DT = np.dtype([
    ("status", "<i4"), ("dp_score", "<i4"), ("score", "<i4"),
    ("q_pos", "<u4"), ("t_pos", "<u4"), ("n_blk", "<u4"),
    ("n_match", "<u4"), ("n_mismatch", "<u4"), ("n_ins", "<u4"), ("n_del", "<u4"),
    ("rows", "<u4"), ("cols", "<u4"), ("ncells", "<i8")], align=True)


def build(force=False):
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("gba_oracle.c", "sdp_oracle.c", "kband_oracle.c", "cigar_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if _ALT is None and not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_ALT or _SO)
        _lib.gbo_align.restype = C.c_int
        _lib.gbo_align_qv.restype = C.c_int
        _lib.gbo_guide_rows.restype = C.c_int
        _lib.gbo_full_dp_score.restype = C.c_int
        _lib.gbo_full_local_score.restype = C.c_int
        _lib.gbo_align_batch.restype = C.c_int
        _lib.gbo_align_batch_mt.restype = C.c_int
        _lib.kbo_align.restype = C.c_int
        _lib.kbo_full_dp.restype = C.c_int
        _lib.sdo_chain.restype = C.c_int
        _lib.sdo_chain_batch.restype = C.c_int
        _lib.cgo_cigar.restype = C.c_int
    return _lib


def make_params(match=-5, mismatch=6, n_score=0, ins=5, dele=4, indel=5, band=10, max_band=250,
                mode=MODE_LINEAR, tie_break=0, affine_open=10, affine_extend=0,
                stats_affine=0, stats_affine_open=0, stats_affine_extend=0, kband_k=0, sub=None, align_type=0):
    p = Params()
    for a in range(5):
        for b in range(5):
            if sub is not None:
                v = int(sub[a][b])
            elif a == 4 or b == 4:
                v = n_score
            else:
                v = match if a == b else mismatch
            p.sub[a][b] = v
    p.ins, p.dele = ins, dele
    p.affine_open, p.affine_extend = affine_open, affine_extend
    p.stats_ins, p.stats_del = indel, indel
    p.stats_affine_open, p.stats_affine_extend = stats_affine_open, stats_affine_extend
    p.band, p.max_band, p.mode, p.tie_break = band, max_band, mode, tie_break
    p.kband_k, p.stats_affine = kband_k, stats_affine
    p.align_type = align_type
    return p


def _u8(x):
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), dtype=np.uint8)
    if isinstance(x, str):
        return np.frombuffer(x.encode(), dtype=np.uint8)
    return np.ascontiguousarray(x, dtype=np.uint8)


def _blocks(blocks):
    if isinstance(blocks, np.ndarray) and blocks.dtype == BLOCK_DT:
        return np.ascontiguousarray(blocks)
    arr = np.zeros(len(blocks), dtype=BLOCK_DT)
    for i, b in enumerate(blocks):
        arr[i] = tuple(int(v) for v in b)
    return arr


def align(q, t, blocks, params, blk_cap=None):
    """Align one pair.  Returns (Result-as-dict, blocks ndarray[BLOCK_DT])."""
    q, t, blk = _u8(q), _u8(t), _blocks(blocks)
    cap = int(blk_cap if blk_cap is not None else max(len(q), 1) + 1)
    out = np.zeros(cap, dtype=BLOCK_DT)
    res = Result()
    lib().gbo_align(q.ctypes.data_as(C.c_void_p), C.c_uint32(len(q)),
                    t.ctypes.data_as(C.c_void_p), C.c_uint32(len(t)),
                    blk.ctypes.data_as(C.c_void_p), C.c_int(len(blk)), C.byref(params),
                    out.ctypes.data_as(C.c_void_p), C.c_uint32(cap), C.byref(res))
    d = {f[0]: getattr(res, f[0]) for f in Result._fields_}
    n = min(d["n_blk"], cap)
    return d, out[:n].copy()


def guide_rows(blocks, q_len, t_len, params):
    """(status, lo[], hi[], ncells) in the clamped 0-based DP coordinates."""
    blk = _blocks(blocks)
    cap = int(q_len) + 2
    lo = np.zeros(cap, dtype=np.int32)
    hi = np.zeros(cap, dtype=np.int32)
    rows = C.c_int32(0)
    ncells = C.c_int64(0)
    st = lib().gbo_guide_rows(blk.ctypes.data_as(C.c_void_p), C.c_int(len(blk)),
                              C.c_uint32(q_len), C.c_uint32(t_len), C.byref(params),
                              lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
                              C.c_int(cap), C.byref(rows), C.byref(ncells))
    return st, lo[:rows.value].copy(), hi[:rows.value].copy(), ncells.value


def full_dp_score(q, t, params):
    q, t = _u8(q), _u8(t)
    s = C.c_int32(0)
    lib().gbo_full_dp_score(q.ctypes.data_as(C.c_void_p), C.c_uint32(len(q)),
                            t.ctypes.data_as(C.c_void_p), C.c_uint32(len(t)),
                            C.byref(params), C.byref(s))
    return s.value


def full_local_score(q, t, params):
    """Unbanded local optimum (Smith-Waterman in lower-is-better scores, linear gaps): the cheapest cell of the whole
    matrix, 0 when nothing scores below zero."""
    q, t = _u8(q), _u8(t)
    out = C.c_int32(0)
    lib().gbo_full_local_score(q.ctypes.data_as(C.c_void_p), C.c_uint32(len(q)), t.ctypes.data_as(C.c_void_p), C.c_uint32(len(t)),
                               C.byref(params), C.byref(out))
    return int(out.value)


def align_batch(batch, params, first=0, count=None, blk_cap_per_pair=None):
    """Run pairs [first, first+count) of a batch dict (keys as blasr_amd.synth.generate
    returns).  Returns (results ndarray[RESULT_DT], list of block arrays)."""
    n = int(batch["n_pairs"])
    count = n - first if count is None else count
    q_len = np.ascontiguousarray(batch["q_len"], dtype=np.uint32)
    caps = (q_len + 1).astype(np.uint32) if blk_cap_per_pair is None else \
        np.full(n, blk_cap_per_pair, dtype=np.uint32)
    out_off = np.zeros(n, dtype=np.uint64)
    out_off[1:] = np.cumsum(caps[:-1], dtype=np.uint64)
    out = np.zeros(int(caps.sum()), dtype=BLOCK_DT)
    results = np.zeros(n, dtype=RESULT_DT)
    assert RESULT_DT.itemsize == C.sizeof(Result)
    arrs = {k: np.ascontiguousarray(batch[k]) for k in
            ("q_arena", "q_off", "t_arena", "t_off", "t_len", "blocks", "blk_off", "n_blk")}
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().gbo_align_batch(p(arrs["q_arena"]), p(arrs["q_off"]), p(q_len),
                          p(arrs["t_arena"]), p(arrs["t_off"]), p(arrs["t_len"]),
                          p(arrs["blocks"]), p(arrs["blk_off"]), p(arrs["n_blk"]),
                          C.byref(params), C.c_uint32(first), C.c_uint32(count),
                          p(results), p(out), p(out_off), p(caps))
    blks = []
    for i in range(first, first + count):
        k = min(int(results[i]["n_blk"]), int(caps[i]))
        blks.append(out[int(out_off[i]):int(out_off[i]) + k].copy())
    return results[first:first + count].copy(), blks


def align_batch_mt(batch, params, threads, seconds=0.0, first=0, count=None):
    """Pairs [first, first+count) on `threads` host threads (pthread workers inside the C
    library, each with its own reusable work space, pulling pairs from a shared counter); stops
    handing out pairs after `seconds` (0: runs them all).  Returns (results, blocks arena,
    out_off, pairs_done, cells_done, elapsed_seconds); pairs are handed out in index order, so
    pairs [first, first+pairs_done) are complete when no time limit cut the run."""
    n = int(batch["n_pairs"])
    count = n - first if count is None else count
    q_len = np.ascontiguousarray(batch["q_len"], dtype=np.uint32)
    caps = (q_len + 1).astype(np.uint32)
    out_off = np.zeros(n, dtype=np.uint64)
    out_off[1:] = np.cumsum(caps[:-1], dtype=np.uint64)
    out = np.zeros(int(caps[first:first + count].sum()) + 1, dtype=BLOCK_DT)
    out_off = out_off - out_off[first] if count else out_off
    results = np.zeros(n, dtype=RESULT_DT)
    arrs = {k: np.ascontiguousarray(batch[k]) for k in
            ("q_arena", "q_off", "t_arena", "t_off", "t_len", "blocks", "blk_off", "n_blk")}
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    pairs_done, cells_done, elapsed = C.c_uint64(0), C.c_uint64(0), C.c_double(0)
    lib().gbo_align_batch_mt(p(arrs["q_arena"]), p(arrs["q_off"]), p(q_len),
                             p(arrs["t_arena"]), p(arrs["t_off"]), p(arrs["t_len"]),
                             p(arrs["blocks"]), p(arrs["blk_off"]), p(arrs["n_blk"]),
                             C.byref(params), C.c_uint32(first), C.c_uint32(count),
                             p(results), p(out), p(out_off), p(caps),
                             C.c_int(int(threads)), C.c_double(float(seconds)),
                             C.byref(pairs_done), C.byref(cells_done), C.byref(elapsed))
    return results, out, out_off, int(pairs_done.value), int(cells_done.value), float(elapsed.value)


# ---- the guide producer (sdp_oracle.c)
SDP_OK, SDP_EMPTY, SDP_TOO_LONG, SDP_TOO_MANY = 0, 1, 7, 8


class SdpParams(C.Structure):
    _fields_ = [("k", C.c_int32), ("ins", C.c_int32), ("dele", C.c_int32), ("match", C.c_int32),
                ("max_fragments", C.c_uint32)]


def make_sdp_params(k=11, ins=5, dele=10, match=5, max_fragments=4096):
    """BLASR's defaults: sdpTupleSize 11, sdpIns 5, sdpDel 10 (iblasr/MappingParameters.h:220-222),
    reward per anchored base = minus the match score of the distance matrix."""
    return SdpParams(k, ins, dele, match, max_fragments)


def sdp_chain(q, t, params, blk_cap=None):
    """One pair.  Returns (status, blocks ndarray[BLOCK_DT], chain_score, n_fragments)."""
    q, t = _u8(q), _u8(t)
    cap = int(blk_cap if blk_cap is not None else params.max_fragments)
    out = np.zeros(max(cap, 1), dtype=BLOCK_DT)
    n_blk, score, n_frag = C.c_uint32(0), C.c_int32(0), C.c_uint32(0)
    st = lib().sdo_chain(q.ctypes.data_as(C.c_void_p), C.c_uint32(len(q)), t.ctypes.data_as(C.c_void_p), C.c_uint32(len(t)),
                         C.byref(params), out.ctypes.data_as(C.c_void_p), C.c_uint32(cap),
                         C.byref(n_blk), C.byref(score), C.byref(n_frag))
    return st, out[:min(n_blk.value, cap)].copy(), score.value, n_frag.value


def sdp_chain_batch(batch, params, first=0, count=None):
    """Pairs [first, first+count) of a batch dict with spelled-out targets.  Returns
    (status, n_blk, chain_score, n_frag, list of block arrays)."""
    n = int(batch["n_pairs"])
    count = n - first if count is None else count
    sl = slice(first, first + count)
    mf = int(params.max_fragments)
    caps = np.full(count, mf, dtype=np.uint32)
    out_off = np.arange(count, dtype=np.uint64) * np.uint64(mf)
    out = np.zeros(max(count * mf, 1), dtype=BLOCK_DT)
    status = np.zeros(count, np.int32); n_blk = np.zeros(count, np.uint32); score = np.zeros(count, np.int32); n_frag = np.zeros(count, np.uint32)
    q_arena = np.ascontiguousarray(batch["q_arena"]); t_arena = np.ascontiguousarray(batch["t_arena"])
    q_off = np.ascontiguousarray(np.asarray(batch["q_off"], dtype=np.uint64)[sl]); t_off = np.ascontiguousarray(np.asarray(batch["t_off"], dtype=np.uint64)[sl])
    q_len = np.ascontiguousarray(np.asarray(batch["q_len"], dtype=np.uint32)[sl]); t_len = np.ascontiguousarray(np.asarray(batch["t_len"], dtype=np.uint32)[sl])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().sdo_chain_batch(p(q_arena), p(q_off), p(q_len), p(t_arena), p(t_off), p(t_len),
                          C.byref(params), C.c_uint32(0), C.c_uint32(count),
                          p(status), p(n_blk), p(score), p(n_frag), p(out), p(out_off), p(caps))
    blks = [out[i * mf:i * mf + int(n_blk[i])].copy() for i in range(count)]
    return status, n_blk, score, n_frag, blks


# ---- fixed-band aligners (kband_oracle.c)
KB_GLOBAL, KB_FIT, KB_EXTEND = 0, 1, 2
KB_LINEAR, KB_AFFINE_HP = 0, 1


class KbandParams(C.Structure):
    _fields_ = [("sub", (C.c_int32 * 5) * 5), ("ins", C.c_int32), ("dele", C.c_int32),
                ("align_type", C.c_int32), ("gap_model", C.c_int32),
                ("hp_ins_open", C.c_int32), ("hp_ins_extend", C.c_int32), ("ins_open", C.c_int32), ("ins_extend", C.c_int32),
                ("hp_rule", C.c_int32), ("tie_break", C.c_int32), ("max_drops", C.c_int32),
                ("stats_ins", C.c_int32), ("stats_del", C.c_int32),
                ("stats_affine", C.c_int32), ("stats_affine_open", C.c_int32), ("stats_affine_extend", C.c_int32)]


def kband_align(q, t, k, params, profile=None, reverse=False, blk_cap=None):
    """One pair through kbo_align.  `params` is any ctypes structure with the layout of KbandParams
    (the library's own gba_kband_params has it).  profile: int16[len(q) * 11] or None.
    Returns (result dict, blocks ndarray[BLOCK_DT])."""
    q, t = _u8(q), _u8(t)
    cap = int(blk_cap if blk_cap is not None else min(len(q), len(t)) + 1)
    out = np.zeros(max(cap, 1), dtype=BLOCK_DT)
    res = Result()
    prof = np.ascontiguousarray(profile, dtype=np.int16) if profile is not None else None
    lib().kbo_align(q.ctypes.data_as(C.c_void_p), C.c_uint32(len(q)), t.ctypes.data_as(C.c_void_p), C.c_uint32(len(t)),
                    C.c_int32(int(k)), C.c_int32(1 if reverse else 0), C.byref(params),
                    prof.ctypes.data_as(C.c_void_p) if prof is not None else None,
                    out.ctypes.data_as(C.c_void_p), C.c_uint32(cap), C.byref(res))
    d = {f[0]: getattr(res, f[0]) for f in Result._fields_}
    return d, out[:min(d["n_blk"], cap)].copy()


def kband_full_dp(q, t, params):
    q, t = _u8(q), _u8(t)
    s = C.c_int32(0)
    lib().kbo_full_dp(q.ctypes.data_as(C.c_void_p), C.c_uint32(len(q)), t.ctypes.data_as(C.c_void_p), C.c_uint32(len(t)),
                      C.byref(params), C.byref(s))
    return s.value


def align_qv(q, t, blocks, params, profile, blk_cap=None):
    """One pair through gbo_align_qv: the guided DP with the costs of a quality-value scorer (profile: int16,
    11 per base of q).  Returns (result dict, blocks)."""
    q, t = _u8(q), _u8(t)
    blk = _blocks(blocks)
    cap = int(blk_cap if blk_cap is not None else len(q) + 1)
    out = np.zeros(max(cap, 1), dtype=BLOCK_DT)
    res = Result()
    prof = np.ascontiguousarray(profile, dtype=np.int16)
    assert len(prof) == 11 * len(q)
    lib().gbo_align_qv(q.ctypes.data_as(C.c_void_p), C.c_uint32(len(q)), t.ctypes.data_as(C.c_void_p), C.c_uint32(len(t)),
                       blk.ctypes.data_as(C.c_void_p), C.c_int(len(blk)), C.byref(params), prof.ctypes.data_as(C.c_void_p),
                       out.ctypes.data_as(C.c_void_p), C.c_uint32(cap), C.byref(res))
    d = {f[0]: getattr(res, f[0]) for f in Result._fields_}
    return d, out[:min(d["n_blk"], cap)].copy()


CIGAR_LETTERS = "MIDNSHP=X"


def cigar_ops(blocks, q, t, q_pos=0, t_pos=0, use_seq_match=True, allow_adjacent_indels=False):
    """cgo_cigar (cigar_oracle.c): the CIGAR of the alignment `blocks` (relative to q_pos / t_pos inside q / t) as a list
    of (count, letter); bases are compared by code (case folded, everything outside ACGT is N)."""
    b = _blocks(blocks)
    qa, ta = _u8(q), _u8(t)
    cap = 2 * (len(qa) + len(ta)) + 4
    ops = np.zeros(cap, np.uint32)
    flags = (1 if use_seq_match else 0) | (2 if allow_adjacent_indels else 0)
    n = lib().cgo_cigar(qa.ctypes.data_as(C.c_void_p), ta.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_int(len(b)),
                        C.c_uint32(int(q_pos)), C.c_uint32(int(t_pos)), C.c_uint32(flags), ops.ctypes.data_as(C.c_void_p), C.c_int(cap))
    if n < 0:
        raise ValueError("blocks overlap or are not increasing")
    return [(int(w) >> 4, CIGAR_LETTERS[int(w) & 15]) for w in ops[:n]]
