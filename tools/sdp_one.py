"""One configuration of gba_sdp_device on one part of the default workload (for counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blasr_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
batch, cfg = synth.preset("c2_10kb_linear", n)
db = api.to_device(batch)
ctx = api.Context(0)
g = ctx.sdp_device(db, api.make_sdp_params())
torch.cuda.synchronize()
print(int((g["status"] == 0).sum()))
