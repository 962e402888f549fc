"""The one-part Local-end-rule leg of the bench line, plain and with a second stream that keeps a few CUs busy with small
kernels all the time: does the traceback behind the last fill (alone on the GPU) run faster when the GPU is not otherwise
idle?  usage: tail_clock.py [pairs = 122880]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # sets GPU_MAX_HW_QUEUES before the runtime starts
import torch
from blasr_amd import api, synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 122880
stop = False


def background(kind):
    s = torch.cuda.Stream()
    x = torch.ones(1 << 14 if kind == "small" else 1 << 26, device="cuda:0")
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(64):
                x.mul_(1.0000001)
            s.synchronize()


for kind in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("none", "small", "none", "large")):
    stop = False
    th = None
    if kind != "none":
        th = threading.Thread(target=background, args=(kind,))
        th.start()
        time.sleep(0.2)
    r = bench.other_workload(torch, api, synth, _lib, "c2_10kb_linear", n, 3, 250, local=True)
    stop = True
    if th:
        th.join()
    print(kind, "ms_per_step", r["ms_per_step"], "fill_ms", r["fill_ms"], "traceback_ms", r["traceback_ms"], "gcups", r["gcups"], flush=True)
