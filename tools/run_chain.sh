#!/bin/bash
# usage: run_chain.sh n   the headline of the bench line with and without GBA_FILL_CHAIN (a part's fills wait for the end
# of the bulk fill of the part before), n times each in turn
for i in $(seq 1 "$1"); do
  for ch in 0 1; do
    GBA_FILL_CHAIN=$ch timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu --no-host --no-guides --no-fixed-band --no-end-to-end --no-others > gpurun_out/chain_${ch}_$i.json 2>gpurun_out/chain_${ch}_$i.err || exit 1
    python3 - "$ch" "$i" <<'P'
import json, sys
d = json.loads(open("gpurun_out/chain_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print("chain", sys.argv[1], "run", sys.argv[2], "gcups", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "fill_ms", d["roofline"]["fill_ms"], "streamed", d["streamed"].get("gcups"))
P
  done
done
