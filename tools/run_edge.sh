#!/bin/bash
# usage: run_edge.sh n v1 v2 ...   the headline of the bench line with GBA_EDGE_PARTS = v (eighths of a part the first part
# of a batch gets; 0: equal parts), n times each in turn
n=$1; shift
for i in $(seq 1 "$n"); do
  for v in "$@"; do
    GBA_EDGE_PARTS=$v timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu --no-host --no-guides --no-fixed-band --no-end-to-end --no-others > gpurun_out/edge_${v}_$i.json 2>gpurun_out/edge_${v}_$i.err || exit 1
    python3 - "$v" "$i" <<'P'
import json, sys
d = json.loads(open("gpurun_out/edge_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print("edge", sys.argv[1], "run", sys.argv[2], "gcups", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "streamed", d["streamed"].get("gcups"), "ok", d["config"]["pairs_ok"])
P
  done
done
