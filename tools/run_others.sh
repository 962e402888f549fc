#!/bin/bash
# usage: run_others.sh name1 name2 ...   other_workloads of the bench line for experiment libraries ("base" = shipped)
for n in "$@"; do
  if [ "$n" = base ]; then unset GBA_LIB; else export GBA_LIB=$PWD/exp_build/libgba_$n.so; fi
  timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-host --no-guides --no-fixed-band --no-end-to-end --no-streamed > gpurun_out/o_$n.json 2>gpurun_out/o_$n.err || exit 1
  python3 - "$n" <<'P'
import json, sys
d = json.loads(open("gpurun_out/o_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], {k: (v["gcups"], v["ms_per_step"], v["fill_ms"], v["traceback_ms"], (v.get("streamed") or {}).get("gcups")) for k, v in d["other_workloads"].items()})
P
done
