"""Times of gba_sdp_device on one part of the default workload, by phase and by tile size (experiments)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blasr_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
batch, cfg = synth.preset("c2_10kb_linear", n)
db = api.to_device(batch)
sp = api.make_sdp_params()
for tile, slots, queue, stop in [(2048, 4, 1024, 2), (2048, 4, 1024, 3), (2048, 4, 1024, 4), (2048, 4, 1024, 0)]:
    os.environ["GBA_SDP_TILE"] = str(tile)
    os.environ["GBA_SDP_SLOTS"] = str(slots)
    os.environ["GBA_SDP_QUEUE"] = str(queue)
    os.environ["GBA_SDP_STOP"] = str(stop)
    ctx = api.Context(0)
    g = ctx.sdp_device(db, sp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g = ctx.sdp_device(db, sp)
    torch.cuda.synchronize()
    print("tile=%d slots=%d queue=%d stop=%d  %.2f ms per %d pairs" % (tile, slots, queue, stop, (time.perf_counter() - t0) / 3 * 1e3, n), flush=True)
    ctx.close()
