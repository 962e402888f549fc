"""Host-side handle over the C-ABI: one Context per (worker, GPU), as include/gba.h asks.

`Context.refine_host` takes numpy batches (gba_submit / gba_wait, H2D + D2H inside;
`Context.submit_host` / `Context.wait_host` are the two halves, `Context.host_array` gives
page-locked numpy arrays from gba_host_alloc);
`Context.refine_device` takes torch CUDA tensors that already live in HBM
(gba_submit_device).  torch is used for device memory and streams only.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BLOCK_DT, PAIR_RESULT_DT, Batch, Config, GbaError, Guides, KbandBatch, KbandParams, Results, Scoring, SdpParams, Timing)


def make_scoring(match=-5, mismatch=6, n_score=0, ins=5, dele=4, indel=5, band=10, max_band=250,
                 mode=_lib.MODE_LINEAR, tie_break=_lib.TIE_DIAG_LEFT_UP, affine_open=10,
                 affine_extend=0, stats_affine=0, stats_affine_open=0, stats_affine_extend=0,
                 kband_k=0, sub=None, extra=(), align_type=_lib.ALIGN_GLOBAL):
    """gba_scoring with BLASR's defaults: SMRTDistanceMatrix-like match/mismatch
    (iblasr/RegisterBlasrOptions.h:358-360), the refine scorer's gap costs
    (iblasr/BlasrUtilsImpl.hpp:630-631 read as ins=5, del=4), rescoring with indel=5 both
    ways (:633-634), guidedAlignBandSize=10 (iblasr/MappingParameters.h:336)."""
    s = Scoring()
    for a in range(5):
        for b in range(5):
            if sub is not None:
                v = int(sub[a][b])
            elif a == 4 or b == 4:
                v = n_score
            else:
                v = match if a == b else mismatch
            s.sub[a][b] = v
    s.ins, s.dele = ins, dele
    s.affine_open, s.affine_extend = affine_open, affine_extend
    s.stats_ins, s.stats_del = indel, indel
    s.stats_affine_open, s.stats_affine_extend = stats_affine_open, stats_affine_extend
    s.band, s.max_band, s.mode, s.tie_break = band, max_band, mode, tie_break
    s.kband_k, s.stats_affine = kband_k, stats_affine
    s.align_type = align_type       # ALIGN_GLOBAL, or ALIGN_LOCAL: free start, end in the band's cheapest cell (linear gaps only)
    # further scorers the finished path is rescored with: dicts with sub (5x5), ins, dele and,
    # for affine gap runs, affine=1, affine_open, affine_extend
    s.n_extra = len(extra)
    for x, e in enumerate(extra):
        r = s.extra[x]
        for a in range(5):
            for b in range(5):
                r.sub[a][b] = int(e["sub"][a][b])
        r.ins, r.dele = int(e.get("ins", 0)), int(e.get("dele", 0))
        r.affine, r.affine_open, r.affine_extend = int(e.get("affine", 0)), int(e.get("affine_open", 0)), int(e.get("affine_extend", 0))
    return s


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_batch_struct(batch):
    """gba_batch over the numpy arrays of a host batch dict.  Returns (Batch, dict of the arrays the
    struct points into — keep it alive as long as the struct is used)."""
    n = int(batch["n_pairs"])
    keep = {k: np.ascontiguousarray(batch[k], dtype=dt) for k, dt in (
        ("q_arena", np.uint8), ("t_arena", np.uint8), ("blocks", BLOCK_DT),
        ("q_off", np.uint64), ("t_off", np.uint64), ("blk_off", np.uint64),
        ("q_len", np.uint32), ("t_len", np.uint32), ("n_blk", np.uint32), ("t_rc", np.uint8))
        if batch.get(k) is not None}
    packed = bool(batch.get("q_packed"))
    if packed and len(keep["q_arena"]) < (int(batch["q_bases"]) + 15) // 16 * 4:
        raise ValueError("packed q_arena shorter than ((q_bases + 15) / 16) * 4 bytes")
    ta = keep.get("t_arena")          # absent: windows of the reference given to set_reference
    bl = keep.get("blocks")           # absent: the guides are made on the device (Context.set_guide_producer)
    b = Batch(n, _ptr(keep["q_arena"]), _ptr(keep["q_off"]), _ptr(keep["q_len"]),
              _ptr(ta) if ta is not None else None, _ptr(keep["t_off"]), _ptr(keep["t_len"]),
              _ptr(bl) if bl is not None else None, _ptr(keep["blk_off"]) if bl is not None else None,
              _ptr(keep["n_blk"]) if bl is not None else None,
              int(batch["q_bases"]) if packed else keep["q_arena"].size, ta.size if ta is not None else 0, bl.size if bl is not None else 0,
              _ptr(keep["t_rc"]) if "t_rc" in keep else None, 1 if packed else 0)
    if batch.get("q_profile") is not None:
        keep["q_profile"] = np.ascontiguousarray(batch["q_profile"], dtype=np.int16)
        if keep["q_profile"].size != 11 * int(b.q_arena_bytes):
            raise ValueError("q_profile must hold 11 costs per base of q_arena")
        b.q_profile = _ptr(keep["q_profile"])
    return b, keep


def pack_reads_2bit(batch):
    """The batch with its reads 2 bits per base (gba_batch.q_packed): q_arena becomes the packed bytes, q_bases the
    number of bases it stands for; offsets and lengths are unchanged.  Raises for a letter outside ACGT / acgt."""
    q = np.ascontiguousarray(batch["q_arena"], dtype=np.uint8)
    code = np.full(256, 255, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
        code[ch + 32] = i
    c = code[q]
    n = len(c)
    bad = np.flatnonzero(c == 255)
    if len(bad):
        # only bytes that belong to a read count (an arena may have unused bytes between its reads): a byte is inside a
        # read when, among the reads that start at or before it, one ends behind it
        off = np.asarray(batch["q_off"], np.int64); end = off + np.asarray(batch["q_len"], np.int64)
        order = np.argsort(off, kind="stable")
        start, reach = off[order], np.maximum.accumulate(end[order])
        idx = np.searchsorted(start, bad, side="right") - 1
        if ((idx >= 0) & (bad < reach[np.maximum(idx, 0)])).any():
            raise ValueError("reads with letters outside ACGT cannot travel packed")
        c[bad] = 0
    if n % 16:
        c = np.concatenate([c, np.zeros(16 - n % 16, np.uint8)])
    c = c.reshape(-1, 4)
    packed = c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)
    out = dict(batch)
    out["q_arena"], out["q_bases"], out["q_packed"] = packed, n, 1
    return out


class Context:
    def __init__(self, device=0, timing=False, stream=None):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        cfg = Config(device, _lib.FLAG_TIMING if timing else 0, stream)
        rc = self._L.gba_create(C.byref(self._h), C.byref(cfg))
        if rc != _lib.OK:
            raise GbaError(rc)
        self.device = device

    def close(self):
        if self._h:
            self._L.gba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=()):
        if rc != _lib.OK and rc not in allow:
            raise GbaError(rc, self._L.gba_last_hip_error(self._h))
        return rc

    def set_scoring(self, scoring):
        self._check(self._L.gba_set_scoring(self._h, C.byref(scoring)))
        self.scoring = scoring

    def set_guide_producer(self, params):
        """gba_set_guide_producer: batches without "blocks" get their guides from the device's own
        chaining of exact matches (SdpParams from make_sdp_params; None switches it off)."""
        self._check(self._L.gba_set_guide_producer(self._h, C.byref(params) if params is not None else None))

    def set_reference(self, ref):
        """Keep `ref` (bytes / uint8 array, the genome as ASCII) in device memory; batches without a
        "t_arena" then describe their targets as windows of it (t_off, t_len, optional t_rc)."""
        if ref is None or len(ref) == 0:
            self._check(self._L.gba_set_reference(self._h, None, 0))
            return
        a = np.ascontiguousarray(np.frombuffer(ref, dtype=np.uint8) if isinstance(ref, (bytes, bytearray)) else ref, dtype=np.uint8)
        self._check(self._L.gba_set_reference(self._h, _ptr(a), a.size))

    def timing(self):
        t = Timing()
        self._check(self._L.gba_get_timing(self._h, C.byref(t)))
        return {f[0]: getattr(t, f[0]) for f in Timing._fields_}

    # ---- host buffers in, host buffers out
    def host_array(self, shape, dtype):
        """A numpy array in page-locked memory of the ctx (gba_host_alloc): batches and result
        buffers built in such arrays travel at the full rate of the link.  Valid until close()."""
        dtype = np.dtype(dtype)
        count = int(np.prod(shape))
        nbytes = max(count * dtype.itemsize, 1)
        p = self._L.gba_host_alloc(self._h, nbytes)
        if not p:
            raise GbaError(_lib.E_NOMEM, self._L.gba_last_hip_error(self._h))
        buf = (C.c_uint8 * nbytes).from_address(p)
        return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)

    def pinned_copy(self, batch):
        """The batch with every array copied into page-locked memory of the ctx."""
        out = {"n_pairs": int(batch["n_pairs"])}
        for k, dt in (("q_arena", np.uint8), ("t_arena", np.uint8), ("blocks", BLOCK_DT), ("q_off", np.uint64),
                      ("t_off", np.uint64), ("blk_off", np.uint64), ("q_len", np.uint32), ("t_len", np.uint32),
                      ("n_blk", np.uint32), ("t_rc", np.uint8)):
            if batch.get(k) is not None:
                a = self.host_array(np.shape(batch[k]), dt)
                a[...] = batch[k]
                out[k] = a
        for k in ("q_packed", "q_bases"):
            if k in batch:
                out[k] = batch[k]
        if batch.get("q_profile") is not None:
            a = self.host_array(np.shape(batch["q_profile"]), np.int16)
            a[...] = batch["q_profile"]
            out["q_profile"] = a
        return out

    def submit_host(self, batch, pairs, blocks, cigar=None):
        """gba_submit: returns at once; `pairs` (ndarray[PAIR_RESULT_DT], one per pair) and `blocks`
        (ndarray[BLOCK_DT]) are filled while the batch runs and must not be read before
        wait_host() has returned.  blocks None (or empty): the blocks are not wanted, only the pair
        results (with the aligned interval) and the CIGARs travel.  cigar: optional (ops ndarray[uint32], off ndarray[uint64] of
        n_pairs + 1, flags) to receive the CIGAR operations as well."""
        b, keep = make_batch_struct(batch)
        if blocks is None:
            blocks = np.zeros(0, dtype=BLOCK_DT)
        r = Results(_ptr(pairs), _ptr(blocks) if len(blocks) else None, len(blocks), 0)
        if cigar is not None:
            ops, off, flags = cigar
            assert ops.dtype == np.uint32 and off.dtype == np.uint64 and len(off) >= int(batch["n_pairs"]) + 1 and len(ops) > 0
            r.cigar_ops, r.cigar_off, r.cigar_capacity, r.cigar_flags = _ptr(ops), _ptr(off), len(ops), int(flags)
        self._check(self._L.gba_submit(self._h, C.byref(b), C.byref(r)))
        self._flight = (b, keep, r, pairs, blocks, cigar)       # everything the running batch points into

    def wait_host(self):
        """gba_wait: blocks until the submitted batch is complete.  Returns (rc, blocks_needed)."""
        rc = self._check(self._L.gba_wait(self._h), allow=(_lib.E_NOSPACE,))
        r = self._flight[2]
        self._flight = None
        self.blocks_needed = int(r.blocks_needed)
        self.cigar_needed = int(r.cigar_needed)
        return rc, self.blocks_needed

    def refine_host(self, batch, blocks_capacity=None):
        """batch: dict of numpy arrays (see blasr_amd.synth.generate).  Returns
        (pairs ndarray[PAIR_RESULT_DT], blocks ndarray[BLOCK_DT], rc)."""
        n = int(batch["n_pairs"])
        if blocks_capacity is None:
            blocks_capacity = int(np.asarray(batch["q_len"]).astype(np.uint64).sum()) + n
        pairs = np.zeros(n, dtype=PAIR_RESULT_DT)
        blocks = np.zeros(int(blocks_capacity), dtype=BLOCK_DT)
        self.submit_host(batch, pairs, blocks)
        rc, needed = self.wait_host()
        return pairs, blocks[:min(needed, int(blocks_capacity))], rc

    def refine_host_intervals(self, batch):
        """The interval (-m 4) form of refine_host: pair results only, each with score, the column
        counts and the aligned interval q_start..q_end / t_start..t_end; no blocks are packed or
        copied out.  Returns (pairs, rc)."""
        pairs = np.zeros(int(batch["n_pairs"]), dtype=PAIR_RESULT_DT)
        self.submit_host(batch, pairs, None)
        rc, _ = self.wait_host()
        return pairs, rc

    def refine_host_cigar(self, batch, flags=_lib.CIGAR_SEQ_MATCH, cigar_capacity=None, blocks_capacity=None):
        """refine_host that also returns the CIGAR operations: (pairs, blocks, ops, off, rc), the
        operations of pair p being ops[off[p]:off[p + 1]] as length << 4 | BAM op code."""
        n = int(batch["n_pairs"])
        qsum = int(np.asarray(batch["q_len"]).astype(np.uint64).sum())
        if blocks_capacity is None:
            blocks_capacity = qsum + n
        if cigar_capacity is None:      # every column its own operation
            cigar_capacity = qsum + int(np.asarray(batch["t_len"]).astype(np.uint64).sum())
        pairs = np.zeros(n, dtype=PAIR_RESULT_DT)
        blocks = np.zeros(int(blocks_capacity), dtype=BLOCK_DT)
        ops = np.zeros(max(int(cigar_capacity), 1), dtype=np.uint32)
        off = np.zeros(n + 1, dtype=np.uint64)
        self.submit_host(batch, pairs, blocks, cigar=(ops, off, flags))
        rc, needed = self.wait_host()
        return pairs, blocks[:min(needed, int(blocks_capacity))], ops[:min(self.cigar_needed, len(ops))], off, rc

    # ---- fixed-band aligners (KBandAlign, AffineKBandAlign, ExtendAlignmentForward / Reverse)
    def kband_host(self, batch, params, blocks_capacity=None):
        """gba_kband on a host batch: dict with q_arena, q_off, q_len, t_arena, t_off, t_len, k (int32 per
        pair) and optionally reverse (uint8 per pair) and profile (int16, 11 per byte of q_arena).
        Returns (pairs ndarray[PAIR_RESULT_DT], blocks ndarray[BLOCK_DT], rc)."""
        n = int(batch["n_pairs"])
        keep = {k: np.ascontiguousarray(batch[k], dtype=dt) for k, dt in
                (("q_arena", np.uint8), ("q_off", np.uint64), ("q_len", np.uint32),
                 ("t_arena", np.uint8), ("t_off", np.uint64), ("t_len", np.uint32), ("k", np.int32))}
        rev = np.ascontiguousarray(batch["reverse"], dtype=np.uint8) if batch.get("reverse") is not None else None
        prof = np.ascontiguousarray(batch["profile"], dtype=np.int16) if batch.get("profile") is not None else None
        if prof is not None and prof.size != keep["q_arena"].size * 11:
            raise ValueError("profile must hold 11 costs per byte of q_arena")
        if blocks_capacity is None:
            blocks_capacity = int(np.minimum(keep["q_len"], keep["t_len"]).astype(np.uint64).sum()) + n
        b = KbandBatch(n, _ptr(keep["q_arena"]), _ptr(keep["q_off"]), _ptr(keep["q_len"]),
                       _ptr(keep["t_arena"]), _ptr(keep["t_off"]), _ptr(keep["t_len"]),
                       keep["q_arena"].size, keep["t_arena"].size, _ptr(keep["k"]),
                       _ptr(rev) if rev is not None else None, _ptr(prof) if prof is not None else None)
        pairs = np.zeros(n, dtype=PAIR_RESULT_DT)
        blocks = np.zeros(max(int(blocks_capacity), 1), dtype=BLOCK_DT)
        r = Results(_ptr(pairs), _ptr(blocks), int(blocks_capacity), 0, None, None, 0, 0, 0)
        rc = self._check(self._L.gba_kband(self._h, C.byref(b), C.byref(params), C.byref(r)), allow=(_lib.E_NOSPACE,))
        return pairs, blocks[:min(int(r.blocks_needed), int(blocks_capacity))], rc

    # ---- guide production
    def sdp_host(self, batch, params, blocks_capacity=None):
        """gba_sdp on a host batch (numpy arrays; its guide entries are not read).  Returns
        (status int32[n], chain_score int32[n], n_blk uint32[n], blk_off uint64[n], blocks
        ndarray[BLOCK_DT], rc)."""
        n = int(batch["n_pairs"])
        if blocks_capacity is None:
            blocks_capacity = max(1, int(np.asarray(batch["q_len"]).astype(np.uint64).sum()) // 8 + n)
        keep = {k: np.ascontiguousarray(batch[k], dtype=dt) for k, dt in
                (("q_arena", np.uint8), ("q_off", np.uint64), ("q_len", np.uint32), ("t_off", np.uint64), ("t_len", np.uint32))}
        for k in ("t_arena", "t_rc"):
            if batch.get(k) is not None:
                keep[k] = np.ascontiguousarray(batch[k], dtype=np.uint8)
        ta, trc = keep.get("t_arena"), keep.get("t_rc")
        packed = bool(batch.get("q_packed"))
        if packed and keep["q_arena"].size < (int(batch["q_bases"]) + 15) // 16 * 4:
            raise ValueError("packed q_arena shorter than ((q_bases + 15) / 16) * 4 bytes")
        b = Batch(n, _ptr(keep["q_arena"]), _ptr(keep["q_off"]), _ptr(keep["q_len"]), _ptr(ta) if ta is not None else None,
                  _ptr(keep["t_off"]), _ptr(keep["t_len"]), None, None, None,
                  int(batch["q_bases"]) if packed else keep["q_arena"].size, ta.size if ta is not None else 0, 0,
                  _ptr(trc) if trc is not None else None, 1 if packed else 0)
        status = np.zeros(n, np.int32); score = np.zeros(n, np.int32); n_blk = np.zeros(n, np.uint32); off = np.zeros(n, np.uint64)
        blocks = np.zeros(int(blocks_capacity), dtype=BLOCK_DT)
        r = Guides(_ptr(status), _ptr(score), _ptr(n_blk), _ptr(off), _ptr(blocks), int(blocks_capacity), 0)
        rc = self._check(self._L.gba_sdp(self._h, C.byref(b), C.byref(params), C.byref(r)), allow=(_lib.E_NOSPACE,))
        return status, score, n_blk, off, blocks[:min(int(r.blocks_needed), int(blocks_capacity))], rc

    def sdp_device(self, dbatch, params, blocks_capacity=None):
        """gba_sdp_device on a device batch (as for refine_device; its "blocks", "blk_off", "n_blk"
        are not read).  Returns a dict of torch tensors on the batch's device: status (int32),
        chain_score (int32), n_blk (int32, bit pattern of uint32), blk_off (int64), blocks (int32
        [capacity, 3]), plus "rc" and "blocks_needed".  n_blk / blk_off / blocks can replace the
        batch's own guide entries for refine_device."""
        import torch
        n = int(dbatch["n_pairs"])
        dev = dbatch["q_arena"].device
        if blocks_capacity is None:
            # a block per ~30 read bases at 15 % error; candidates of one read share its bytes, so count lengths, not the arena
            blocks_capacity = max(1, int(dbatch["q_len"].to(torch.int64).sum().item()) // 8 + n)
        g = {"status": torch.zeros(n, dtype=torch.int32, device=dev), "chain_score": torch.zeros(n, dtype=torch.int32, device=dev),
             "n_blk": torch.zeros(n, dtype=torch.int32, device=dev), "blk_off": torch.zeros(n, dtype=torch.int64, device=dev),
             "blocks": torch.zeros((int(blocks_capacity), 3), dtype=torch.int32, device=dev)}
        ta = dbatch.get("t_arena")
        trc = dbatch.get("t_rc")
        packed = bool(dbatch.get("q_packed"))
        b = Batch(n, dbatch["q_arena"].data_ptr(), dbatch["q_off"].data_ptr(), dbatch["q_len"].data_ptr(),
                  ta.data_ptr() if ta is not None else None, dbatch["t_off"].data_ptr(), dbatch["t_len"].data_ptr(),
                  None, None, None, int(dbatch["q_bases"]) if packed else dbatch["q_arena"].numel(), ta.numel() if ta is not None else 0, 0,
                  trc.data_ptr() if trc is not None else None, 1 if packed else 0)
        r = Guides(g["status"].data_ptr(), g["chain_score"].data_ptr(), g["n_blk"].data_ptr(), g["blk_off"].data_ptr(),
                   g["blocks"].data_ptr(), int(blocks_capacity), 0)
        g["rc"] = self._check(self._L.gba_sdp_device(self._h, C.byref(b), C.byref(params), C.byref(r)), allow=(_lib.E_NOSPACE,))
        g["blocks_needed"] = int(r.blocks_needed)
        return g

    # ---- device-resident tensors
    def refine_device(self, dbatch, out_pairs, out_blocks, no_alloc=False, cigar=None):
        """dbatch: dict of torch CUDA tensors with the same keys as a host batch (uint8 arenas;
        offsets/lengths as int64/int32 tensors holding the uint64/uint32 bit patterns; blocks as
        an int32 tensor [n_blocks, 3]).  out_pairs: uint8 tensor of n*PAIR_RESULT_DT.itemsize
        bytes; out_blocks: int32 tensor [capacity, 3], or None when the blocks are not wanted (pair
        results with the aligned interval only).  Returns (rc, blocks_needed).
        no_alloc: through gba_submit_device_noalloc (fixed scratch, no device allocation; a batch
        that does not fit raises GbaError with code E_TOOBIG).
        cigar: optional (ops int32 tensor [capacity], off int64 tensor [n + 1], flags); the total number
        of operations is left in self.cigar_needed."""
        n = int(dbatch["n_pairs"])
        ta = dbatch.get("t_arena")
        trc = dbatch.get("t_rc")
        gb = dbatch.get("blocks")         # absent: guides made on the device (set_guide_producer)
        packed = bool(dbatch.get("q_packed"))     # reads 2 bits per base, q_bases of them (pack_reads_2bit)
        if packed and dbatch["q_arena"].numel() < (int(dbatch["q_bases"]) + 15) // 16 * 4:
            raise ValueError("packed q_arena shorter than ((q_bases + 15) / 16) * 4 bytes")
        b = Batch(n, dbatch["q_arena"].data_ptr(), dbatch["q_off"].data_ptr(), dbatch["q_len"].data_ptr(),
                  ta.data_ptr() if ta is not None else None, dbatch["t_off"].data_ptr(), dbatch["t_len"].data_ptr(),
                  gb.data_ptr() if gb is not None else None, dbatch["blk_off"].data_ptr() if gb is not None else None,
                  dbatch["n_blk"].data_ptr() if gb is not None else None,
                  int(dbatch["q_bases"]) if packed else dbatch["q_arena"].numel(), ta.numel() if ta is not None else 0,
                  gb.shape[0] if gb is not None else 0, trc.data_ptr() if trc is not None else None, 1 if packed else 0)
        if dbatch.get("q_profile") is not None:       # int16 tensor, 11 per base of q_arena
            assert dbatch["q_profile"].element_size() == 2 and dbatch["q_profile"].numel() == 11 * int(b.q_arena_bytes)
            b.q_profile = dbatch["q_profile"].data_ptr()
        assert out_pairs.numel() * out_pairs.element_size() >= n * PAIR_RESULT_DT.itemsize
        r = Results(out_pairs.data_ptr(), out_blocks.data_ptr() if out_blocks is not None else None,
                    out_blocks.shape[0] if out_blocks is not None else 0, 0)
        if cigar is not None:
            ops, off, flags = cigar
            assert ops.element_size() == 4 and off.element_size() == 8 and off.numel() >= n + 1 and ops.numel() > 0
            r.cigar_ops, r.cigar_off, r.cigar_capacity, r.cigar_flags = ops.data_ptr(), off.data_ptr(), ops.numel(), int(flags)
        fn = self._L.gba_submit_device_noalloc if no_alloc else self._L.gba_submit_device
        rc = self._check(fn(self._h, C.byref(b), C.byref(r)), allow=(_lib.E_NOSPACE,))
        self.cigar_needed = int(r.cigar_needed)
        return rc, int(r.blocks_needed)


def make_kband_params(align_type=_lib.KB_GLOBAL, gap_model=_lib.KB_LINEAR, match=-5, mismatch=6, n_score=0, ins=5, dele=5,
                      indel=5, sub=None, hp_rule=0, tie_break=_lib.TIE_DIAG_LEFT_UP, max_drops=10,
                      hp_ins_open=None, hp_ins_extend=None, ins_open=None, ins_extend=None,
                      stats_affine=0, stats_affine_open=0, stats_affine_extend=0):
    """gba_kband_params.  The affine costs default to what PairwiseLocalAlign passes
    (iblasr/BlasrAlignImpl.hpp:1259-1262): homopolymer insertion indel + 2 / indel - 3, any insertion
    indel + 2 / indel - 1, deletion indel; max_drops to params.maxExtendDropoff
    (iblasr/MappingParameters.h:288); rescoring with (indel, indel)."""
    p = KbandParams()
    for a in range(5):
        for b in range(5):
            p.sub[a][b] = int(sub[a][b]) if sub is not None else (n_score if a == 4 or b == 4 else (match if a == b else mismatch))
    p.ins, p.dele, p.align_type, p.gap_model = ins, dele, align_type, gap_model
    p.hp_ins_open = indel + 2 if hp_ins_open is None else hp_ins_open
    p.hp_ins_extend = indel - 3 if hp_ins_extend is None else hp_ins_extend
    p.ins_open = indel + 2 if ins_open is None else ins_open
    p.ins_extend = indel - 1 if ins_extend is None else ins_extend
    p.hp_rule, p.tie_break, p.max_drops = hp_rule, tie_break, max_drops
    p.stats_ins, p.stats_del = indel, indel
    p.stats_affine, p.stats_affine_open, p.stats_affine_extend = stats_affine, stats_affine_open, stats_affine_extend
    return p


def make_sdp_params(tuple_size=11, ins=5, dele=10, match=5, max_fragments=1024):
    """gba_sdp_params with BLASR's defaults: sdpTupleSize 11, sdpIns 5, sdpDel 10
    (iblasr/MappingParameters.h:220-222), a reward of 5 per anchored base (minus the match score)."""
    return SdpParams(tuple_size, ins, dele, match, max_fragments)


class HostPipeline:
    """Several contexts on ONE GPU used in turn, so that host-buffer batches overlap: batch k+1 is
    submitted before batch k is waited for, its inputs travel while batch k computes, and batch k's
    last results travel while batch k+1 starts.  This is what the reference's worker threads become
    (one MappingBuffers = one context per thread, Blasr.cpp:771): with two or more of them per GPU
    the link and the GPU are both kept busy.  Set GPU_MAX_HW_QUEUES=16 in the environment before the
    HIP runtime starts (8 streams per context).

    Batches that start together stay together: both wait for their first inputs at the same time and
    both copy their last results out at the same time, with the GPU idle in between (measured: 188 ms
    per batch in step against 162 ms half a batch apart, for 491 520 pairs of 10 kb).  So a batch
    submitted while another one has been running for less than `stagger` of a batch's usual time is
    held back inside submit() for the difference; once apart the batches stay apart, and nothing is
    held back any more."""

    def __init__(self, device=0, depth=2, stagger=0.4):
        import time
        self._now = time.perf_counter
        self._sleep = time.sleep
        self.ctxs = [Context(device) for _ in range(depth)]
        self._sub = self._wait = 0
        self.stagger = stagger
        self._service = None          # usual time of a batch: from its submit, or the completion before it, to its completion
        self._t_submit = []           # submit times of the batches in flight, oldest first
        self._t_done = None

    def close(self):
        for c in self.ctxs:
            c.close()

    def set_scoring(self, scoring):
        for c in self.ctxs:
            c.set_scoring(scoring)

    def set_guide_producer(self, params):
        for c in self.ctxs:
            c.set_guide_producer(params)

    def set_reference(self, ref):
        for c in self.ctxs:
            c.set_reference(ref)

    def host_array(self, shape, dtype):
        return self.ctxs[0].host_array(shape, dtype)

    def pinned_copy(self, batch):
        return self.ctxs[0].pinned_copy(batch)

    def submit(self, batch, pairs, blocks, cigar=None):
        if self._sub - self._wait >= len(self.ctxs):
            raise GbaError(_lib.E_STATE)
        if self._t_submit and self._service and self.stagger > 0:
            hold = self._t_submit[-1] + self.stagger * self._service - self._now()
            if hold > 0:
                self._sleep(hold)
        self.ctxs[self._sub % len(self.ctxs)].submit_host(batch, pairs, blocks, cigar=cigar)
        self._t_submit.append(self._now())
        self._sub += 1

    def wait(self):
        """Waits for the oldest submitted batch.  Returns (rc, blocks_needed)."""
        if self._wait >= self._sub:
            raise GbaError(_lib.E_STATE)
        r = self.ctxs[self._wait % len(self.ctxs)].wait_host()
        self._wait += 1
        now, started = self._now(), self._t_submit.pop(0)
        took = now - (max(started, self._t_done) if self._t_done is not None else started)
        self._service = took if self._service is None else 0.5 * self._service + 0.5 * took
        self._t_done = now
        return r

    def rerun_last(self, batch, pairs, blocks, cigar=None):
        """Runs a batch to completion on the context whose batch wait() returned last (it is idle; the batches still in
        flight are on the others): for a caller that has to repeat that batch, e.g. with a larger block arena after
        GBA_E_NOSPACE.  Returns (rc, blocks_needed)."""
        if self._wait == 0:
            raise GbaError(_lib.E_STATE)
        c = self.ctxs[(self._wait - 1) % len(self.ctxs)]
        c.submit_host(batch, pairs, blocks, cigar=cigar)
        return c.wait_host()


class DevicePipeline:
    """HostPipeline's counterpart for batches that are already in device memory: several contexts on ONE GPU used in turn,
    a worker thread each (gba_submit_device returns when its batch has completed), so that batch k+1 starts while batch k
    runs: the first packing and guide of one batch and the last traceback of the other run under fills instead of alone,
    and a batch of a single part (too few pairs to be cut) no longer has its traceback to itself.  Measured: 491 520
    pairs of 10 kb 830 -> 890-903 GCUPS, 16 384 affine pairs of 20 kb 193 -> 256-262.  Every context holds its own scratch.
    Set GPU_MAX_HW_QUEUES=20 in the environment before the HIP runtime starts (two contexts' kernel
    streams then have queues of their own; 24 and more slowed every later context of the process down)."""

    def __init__(self, device=0, depth=2, contexts=()):
        """contexts: contexts of this GPU the caller already has (they keep their scoring, reference and scratch and stay
        open when the pipeline is closed); the pipeline adds its own up to `depth`."""
        from concurrent.futures import ThreadPoolExecutor
        self._own = [Context(device) for _ in range(max(depth - len(contexts), 0))]
        self.ctxs = list(contexts) + self._own
        self._pools = [ThreadPoolExecutor(max_workers=1) for _ in self.ctxs]
        self._futures = []            # batches in flight, oldest first
        self._sub = 0

    def close(self):
        for f in self._futures:
            f.exception()             # let what is in flight finish
        self._futures = []
        for p in self._pools:
            p.shutdown(wait=True)
        for c in self._own:
            c.close()

    def set_scoring(self, scoring):
        for c in self.ctxs:
            c.set_scoring(scoring)

    def set_guide_producer(self, params):
        for c in self.ctxs:
            c.set_guide_producer(params)

    def set_reference(self, ref):
        for c in self.ctxs:
            c.set_reference(ref)

    def submit(self, dbatch, out_pairs, out_blocks, cigar=None):
        """Queues a batch on the next context (arguments as Context.refine_device).  At most `depth` batches may be in
        flight; their inputs and outputs must stay untouched until wait() has returned them."""
        if len(self._futures) >= len(self.ctxs):
            raise GbaError(_lib.E_STATE)
        i = self._sub % len(self.ctxs)
        self._futures.append(self._pools[i].submit(self.ctxs[i].refine_device, dbatch, out_pairs, out_blocks, False, cigar))
        self._sub += 1

    def wait(self):
        """Waits for the oldest submitted batch.  Returns (rc, blocks_needed); raises what its call raised."""
        if not self._futures:
            raise GbaError(_lib.E_STATE)
        return self._futures.pop(0).result()


def to_device(batch, device="cuda:0"):
    """Copy a host batch dict to torch CUDA tensors (bit patterns preserved)."""
    import torch
    d = {"n_pairs": int(batch["n_pairs"])}
    for k in ("q_arena", "t_arena", "t_rc"):
        if batch.get(k) is not None:
            d[k] = torch.from_numpy(np.ascontiguousarray(batch[k], dtype=np.uint8)).to(device)
    guides = batch.get("blocks") is not None      # absent: sequences only (guides made on the device)
    for k in ("q_off", "t_off") + (("blk_off",) if guides else ()):
        d[k] = torch.from_numpy(np.ascontiguousarray(batch[k], dtype=np.uint64).view(np.int64)).to(device)
    for k in ("q_len", "t_len") + (("n_blk",) if guides else ()):
        d[k] = torch.from_numpy(np.ascontiguousarray(batch[k], dtype=np.uint32).view(np.int32)).to(device)
    if guides:
        blk = np.ascontiguousarray(batch["blocks"], dtype=BLOCK_DT)
        d["blocks"] = torch.from_numpy(blk.view(np.uint32).reshape(-1, 3).view(np.int32)).to(device)
    if batch.get("q_profile") is not None:
        d["q_profile"] = torch.from_numpy(np.ascontiguousarray(batch["q_profile"], dtype=np.int16)).to(device)
    if batch.get("q_packed"):
        d["q_packed"], d["q_bases"] = 1, int(batch["q_bases"])
    return d
