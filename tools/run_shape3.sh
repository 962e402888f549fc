#!/bin/bash
# usage: run_shape3.sh n shape ...   headline and streamed leg of the bench line with GBA_PART_SHAPE = shape ("default": the library's own parts), n times each in turn
n=$1; shift
for i in $(seq 1 "$n"); do
  for sh in "$@"; do
    if [ "$sh" = default ]; then unset GBA_PART_SHAPE; else export GBA_PART_SHAPE=$sh; fi
    timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu --no-host --no-guides --no-fixed-band --no-end-to-end --no-others > gpurun_out/shape3.json 2>gpurun_out/shape3.err || exit 1
    python3 - "$sh" "$i" <<'P'
import json, sys
d = json.loads(open("gpurun_out/shape3.json").read().strip().splitlines()[-1])
print("shape", sys.argv[1], "run", sys.argv[2], "gcups", d["value"], "streamed", d["streamed"].get("gcups"), d["streamed"].get("ms_per_batch"))
P
  done
done
