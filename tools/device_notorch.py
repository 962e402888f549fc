"""The device-resident step without PyTorch in the process (the system's HIP runtime, what a C++ caller links): device buffers
through hipMalloc / hipMemcpy over ctypes, then gba_submit_device as bench.py's headline does.  usage: device_notorch.py [pairs] [steps] [workload] [band]"""
import ctypes as C, os, sys, time
os.environ.setdefault("GBA_NO_TORCH", "1")
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("NOTORCH_HW_QUEUES", "16")      # (the GPU boxes export HIP's default, 4)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from blasr_amd import api, synth, _lib

hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
if os.environ.get("NOTORCH_DEVICE_FLAGS"):          # experiments: hipSetDeviceFlags before the first HIP call (1: spin, 2: yield, 4: blocking sync)
    print("hipSetDeviceFlags", hip.hipSetDeviceFlags(C.c_uint(int(os.environ["NOTORCH_DEVICE_FLAGS"]))))


class Dev:
    """The few tensor methods api.Context.refine_device uses, over a hipMalloc'ed buffer."""
    def __init__(self, arr=None, nbytes=0, shape=None, itemsize=1):
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            nbytes, shape, itemsize = arr.nbytes, arr.shape, arr.itemsize
        self.shape, self._n, self._item = shape, nbytes // itemsize, itemsize
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        self.ptr = p.value
        if arr is not None:
            assert hip.hipMemcpy(self.ptr, arr.ctypes.data, nbytes, 1) == 0
        else:
            assert hip.hipMemset(self.ptr, 0, nbytes) == 0
    def data_ptr(self): return self.ptr
    def numel(self): return self._n
    def element_size(self): return self._item
    def host(self, dtype, count):
        out = np.empty(count, dtype)
        assert hip.hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2) == 0
        return out


n = int(sys.argv[1]) if len(sys.argv) > 1 else 491520
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wl = sys.argv[3] if len(sys.argv) > 3 else "c2_10kb_linear"
cfg = dict(synth.PRESETS[wl])
if len(sys.argv) > 4:
    cfg["band"] = int(sys.argv[4])
k = cfg.get("decoy_every", 0) or 1
ref = synth.reference(cfg["ref_len"], synth.REF_SEED, cfg.get("n_frac", 0.0))
host = synth.generate(n, share_reads=k > 1, ref=ref, **{a: b for a, b in cfg.items() if a != "ref"})
d = {"n_pairs": n}
for k in ("q_arena", "t_arena", "blocks", "q_off", "t_off", "blk_off", "q_len", "t_len", "n_blk", "t_rc"):
    if host.get(k) is not None:
        a = np.asarray(host[k])
        d[k] = Dev(a.view(np.uint32).reshape(-1, 3) if k == "blocks" else a)
cap = int(host["q_len"].astype(np.int64).sum() // 4) + n
out_pairs = Dev(nbytes=n * _lib.PAIR_RESULT_DT.itemsize)
out_blocks = Dev(nbytes=cap * 12, shape=(cap, 3), itemsize=4)
ctx = api.Context(0, timing=True)
ctx.set_scoring(api.make_scoring(band=cfg["band"], mode=_lib.MODE_AFFINE if cfg["mode"] == "affine" else _lib.MODE_LINEAR))
ctx.refine_device(d, out_pairs, out_blocks)
t0 = time.perf_counter()
for _ in range(steps):
    rc, needed = ctx.refine_device(d, out_pairs, out_blocks)
dt = (time.perf_counter() - t0) / steps
tm = ctx.timing()
r = out_pairs.host(np.uint8, n * _lib.PAIR_RESULT_DT.itemsize).view(_lib.PAIR_RESULT_DT)
cells = int(r["ncells"][r["status"] == 0].sum())
print(wl + ", no PyTorch in the process: %d pairs, %.1f ms per step, %.1f GCUPS, ok %d, rc %d, fill in flight %.1f ms" % (n, dt * 1e3, cells / dt / 1e9, int((r["status"] == 0).sum()), rc, tm["fill_busy_ms"]))
