"""gba_sdp_device on one part of the default workload with the library GBA_LIB names (experiments)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blasr_amd import api, synth
n = 32768
batch, cfg = synth.preset("c2_10kb_linear", n)
db = api.to_device(batch)
sp = api.make_sdp_params()
for tile, slots in [(2048, 4), (2048, 8), (1024, 8), (1024, 16), (512, 16)]:
    os.environ["GBA_SDP_TILE"] = str(tile); os.environ["GBA_SDP_SLOTS"] = str(slots)
    for stop in (2,):
        os.environ["GBA_SDP_STOP"] = str(stop)
        ctx = api.Context(0)
        g = ctx.sdp_device(db, sp); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            g = ctx.sdp_device(db, sp)
        torch.cuda.synchronize()
        print("%s tile=%d slots=%d stop=%d  %.2f ms" % (os.path.basename(os.environ.get("GBA_LIB", "libgba.so")), tile, slots, stop, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
        ctx.close()
