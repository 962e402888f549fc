#!/bin/bash
# usage: run_shape2.sh workload pairs shape ...   one workload of bench.py with GBA_PART_SHAPE = shape ("default": the library's own parts)
wl=$1; pairs=$2; shift; shift
for sh in "$@"; do
  if [ "$sh" = default ]; then unset GBA_PART_SHAPE; else export GBA_PART_SHAPE=$sh; fi
  timeout -k 10 200 python3 bench.py --workload $wl --pairs $pairs --steps 4 --warmup 1 --no-cpu --no-host --no-guides --no-fixed-band --no-end-to-end --no-others --no-streamed > gpurun_out/shape2.json 2>gpurun_out/shape2.err || exit 1
  python3 - "$wl" "$pairs" "$sh" <<'P'
import json, sys
d = json.loads(open("gpurun_out/shape2.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "shape", sys.argv[3], "gcups", d["value"], "ms", d["ms_per_step"], "fill_ms", d["roofline"]["fill_ms"], "tb", d["traceback_ms"])
P
done
