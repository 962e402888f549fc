#!/bin/bash
# usage: run_edge2.sh n edge:chain ...   headline with GBA_EDGE_PARTS / GBA_FILL_CHAIN combinations, n times each in turn
n=$1; shift
for i in $(seq 1 "$n"); do
  for combo in "$@"; do
    v=${combo%%:*}; ch=${combo##*:}
    GBA_EDGE_PARTS=$v GBA_FILL_CHAIN=$ch timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu --no-host --no-guides --no-fixed-band --no-end-to-end --no-others > gpurun_out/edge2_${v}_${ch}_$i.json 2>gpurun_out/edge2_${v}_${ch}_$i.err || exit 1
    python3 - "$v" "$ch" "$i" <<'P'
import json, sys
d = json.loads(open("gpurun_out/edge2_%s_%s_%s.json" % tuple(sys.argv[1:4])).read().strip().splitlines()[-1])
print("edge", sys.argv[1], "chain", sys.argv[2], "run", sys.argv[3], "gcups", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "streamed", d["streamed"].get("gcups"))
P
  done
done
