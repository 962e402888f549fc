"""ctypes binding of include/gba.h (libgba.so, built in-tree by blasr_amd.build).

There is no CPU fallback: if the HIP library is missing or fails to load, every
entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GBA_LIB", os.path.join(_HERE, "libgba.so"))   # GBA_LIB: experiment builds

# return codes / statuses / modes (include/gba.h)
OK, E_INVALID, E_NOMEM, E_HIP, E_NOSPACE, E_UNSUPPORTED, E_STATE, E_TOOBIG = 0, -1, -2, -3, -4, -5, -6, -7
NOALLOC_SCRATCH_BYTES = 64 << 20
(PAIR_OK, PAIR_EMPTY_GUIDE, PAIR_BAD_GUIDE, PAIR_UNREACHABLE, PAIR_NOSPACE,
 PAIR_END_OUT_OF_BAND, PAIR_TOO_WIDE, PAIR_TOO_LONG, PAIR_TOO_MANY) = range(9)
MODE_LINEAR, MODE_AFFINE, MODE_KBAND = 0, 1, 2
TIE_DIAG_LEFT_UP, TIE_DIAG_UP_LEFT = 0, 1
ALIGN_GLOBAL, ALIGN_LOCAL = 0, 1
PAIR_NO_ALIGNMENT = 9
FLAG_TIMING = 1
CIGAR_SEQ_MATCH, CIGAR_ADJACENT_INDELS = 1, 2
CIGAR_OPS = "MIDNSHP=X"

BLOCK_DT = np.dtype([("q_pos", "<u4"), ("t_pos", "<u4"), ("length", "<u4")])
PAIR_RESULT_DT = np.dtype([
    ("status", "<i4"), ("dp_score", "<i4"), ("score", "<i4"),
    ("q_pos", "<u4"), ("t_pos", "<u4"), ("n_blk", "<u4"), ("blk_off", "<u8"),
    ("n_match", "<u4"), ("n_mismatch", "<u4"), ("n_ins", "<u4"), ("n_del", "<u4"),
    ("rows", "<u4"), ("cols", "<u4"), ("ncells", "<u8"), ("extra_score", "<i4", (2,)),
    ("q_start", "<u4"), ("q_end", "<u4"), ("t_start", "<u4"), ("t_end", "<u4")], align=True)


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32), ("stream", C.c_void_p)]


class Rescorer(C.Structure):
    _fields_ = [("sub", (C.c_int32 * 5) * 5), ("ins", C.c_int32), ("dele", C.c_int32),
                ("affine_open", C.c_int32), ("affine_extend", C.c_int32), ("affine", C.c_int32)]


MAX_EXTRA = 2


class Scoring(C.Structure):
    _fields_ = [
        ("sub", (C.c_int32 * 5) * 5),
        ("ins", C.c_int32), ("dele", C.c_int32),
        ("affine_open", C.c_int32), ("affine_extend", C.c_int32),
        ("stats_ins", C.c_int32), ("stats_del", C.c_int32),
        ("stats_affine_open", C.c_int32), ("stats_affine_extend", C.c_int32),
        ("band", C.c_int32), ("max_band", C.c_int32),
        ("mode", C.c_int32), ("tie_break", C.c_int32),
        ("kband_k", C.c_int32), ("stats_affine", C.c_int32),
        ("n_extra", C.c_int32), ("extra", Rescorer * MAX_EXTRA), ("align_type", C.c_int32),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("n_pairs", C.c_uint32),
        ("q_arena", C.c_void_p), ("q_off", C.c_void_p), ("q_len", C.c_void_p),
        ("t_arena", C.c_void_p), ("t_off", C.c_void_p), ("t_len", C.c_void_p),
        ("blocks", C.c_void_p), ("blk_off", C.c_void_p), ("n_blk", C.c_void_p),
        ("q_arena_bytes", C.c_uint64), ("t_arena_bytes", C.c_uint64), ("n_blocks_total", C.c_uint64),
        ("t_rc", C.c_void_p), ("q_packed", C.c_uint32), ("q_profile", C.c_void_p),
    ]


class Results(C.Structure):
    _fields_ = [("pairs", C.c_void_p), ("blocks", C.c_void_p),
                ("blocks_capacity", C.c_uint64), ("blocks_needed", C.c_uint64),
                ("cigar_ops", C.c_void_p), ("cigar_off", C.c_void_p),
                ("cigar_capacity", C.c_uint64), ("cigar_needed", C.c_uint64), ("cigar_flags", C.c_uint32)]


class Timing(C.Structure):
    _fields_ = [("fill_ms", C.c_float), ("traceback_ms", C.c_float), ("total_ms", C.c_float),
                ("fill_launches", C.c_uint32), ("cells", C.c_uint64), ("rows", C.c_uint64),
                ("record_bytes", C.c_uint64), ("wide_pairs", C.c_uint32), ("scratch_retries", C.c_uint32),
                ("device_bytes", C.c_uint64), ("slot_bytes", C.c_uint64), ("fill_busy_ms", C.c_float)]


class SdpParams(C.Structure):
    _fields_ = [("tuple_size", C.c_int32), ("ins", C.c_int32), ("dele", C.c_int32), ("match", C.c_int32),
                ("max_fragments", C.c_uint32)]


class Guides(C.Structure):
    _fields_ = [("status", C.c_void_p), ("chain_score", C.c_void_p), ("n_blk", C.c_void_p), ("blk_off", C.c_void_p),
                ("blocks", C.c_void_p), ("blocks_capacity", C.c_uint64), ("blocks_needed", C.c_uint64)]


KB_GLOBAL, KB_FIT, KB_EXTEND = 0, 1, 2
KB_LINEAR, KB_AFFINE_HP = 0, 1


class KbandParams(C.Structure):
    _fields_ = [("sub", (C.c_int32 * 5) * 5), ("ins", C.c_int32), ("dele", C.c_int32),
                ("align_type", C.c_int32), ("gap_model", C.c_int32),
                ("hp_ins_open", C.c_int32), ("hp_ins_extend", C.c_int32), ("ins_open", C.c_int32), ("ins_extend", C.c_int32),
                ("hp_rule", C.c_int32), ("tie_break", C.c_int32), ("max_drops", C.c_int32),
                ("stats_ins", C.c_int32), ("stats_del", C.c_int32),
                ("stats_affine", C.c_int32), ("stats_affine_open", C.c_int32), ("stats_affine_extend", C.c_int32)]


class KbandBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_uint32),
                ("q_arena", C.c_void_p), ("q_off", C.c_void_p), ("q_len", C.c_void_p),
                ("t_arena", C.c_void_p), ("t_off", C.c_void_p), ("t_len", C.c_void_p),
                ("q_arena_bytes", C.c_uint64), ("t_arena_bytes", C.c_uint64),
                ("k", C.c_void_p), ("reverse", C.c_void_p), ("profile", C.c_void_p)]


EXPORTS = ("gba_kband", "gba_create", "gba_destroy", "gba_set_scoring", "gba_submit", "gba_wait",
           "gba_submit_device", "gba_submit_device_noalloc", "gba_sdp_device", "gba_sdp", "gba_set_guide_producer", "gba_host_alloc", "gba_host_free", "gba_set_reference", "gba_get_timing", "gba_last_hip_error", "gba_strerror", "gba_version")

_lib = None


def lib():
    """Load libgba.so; raises RuntimeError when it is missing (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgba.so not built: run `python -m blasr_amd.build` (hipcc, gfx950)")
        # One HIP runtime per process: PyTorch ships its own libamdhip64; importing it first makes
        # libgba.so bind to that copy, so tensors and gba contexts share one runtime.
        # (GBA_NO_TORCH=1: a process that only uses host buffers needs no PyTorch and runs on the
        # system's HIP runtime.)
        if not os.environ.get("GBA_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(L, name)          # AttributeError if a declared symbol is not exported
        L.gba_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Config)]
        L.gba_destroy.argtypes = [C.c_void_p]
        L.gba_destroy.restype = None
        L.gba_set_scoring.argtypes = [C.c_void_p, C.POINTER(Scoring)]
        L.gba_submit.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Results)]
        L.gba_wait.argtypes = [C.c_void_p]
        L.gba_host_alloc.argtypes = [C.c_void_p, C.c_uint64]
        L.gba_host_alloc.restype = C.c_void_p
        L.gba_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.gba_host_free.restype = None
        L.gba_submit_device.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Results)]
        L.gba_submit_device_noalloc.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Results)]
        L.gba_sdp_device.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(SdpParams), C.POINTER(Guides)]
        L.gba_sdp.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(SdpParams), C.POINTER(Guides)]
        L.gba_set_guide_producer.argtypes = [C.c_void_p, C.POINTER(SdpParams)]
        L.gba_set_reference.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.gba_kband.argtypes = [C.c_void_p, C.POINTER(KbandBatch), C.POINTER(KbandParams), C.POINTER(Results)]
        L.gba_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        L.gba_last_hip_error.argtypes = [C.c_void_p]
        L.gba_strerror.argtypes = [C.c_int]
        L.gba_strerror.restype = C.c_char_p
        _lib = L
    return _lib


class GbaError(RuntimeError):
    def __init__(self, code, hip=0):
        self.code = code
        msg = lib().gba_strerror(code).decode()
        if hip:
            msg += " (hipError %d)" % hip
        super().__init__("gba: %s [%d]" % (msg, code))
