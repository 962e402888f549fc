#!/bin/bash
# affine workload: the third lane class (slots) against none, at two batch sizes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for n in 98304 49152; do for top in 0 96 80 128; do
  GBA_AFFINE_TOP_SLOTS=$top timeout -k 10 200 python bench.py --workload c3_20kb_affine --pairs $n --steps 2 --warmup 1 --no-cpu --no-host --no-guides > gpurun_out/c3top.json 2> gpurun_out/c3top.err || { tail -5 gpurun_out/c3top.err; exit 1; }
  python - <<P
import json
d = json.loads(open("gpurun_out/c3top.json").read().strip().splitlines()[-1])
print("pairs $n top $top  %.1f GCUPS  %.1f ms  fill %.1f ms  tb %.1f ms  ok %d" % (d["value"], d["ms_per_step"], d["roofline"]["fill_ms"], d["traceback_ms"], d["config"]["pairs_ok"]))
P
done; done
