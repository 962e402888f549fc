#!/bin/bash
# usage: run_queues.sh q1 q2 ...   headline, streamed and other_workloads of the bench line at different hardware queue counts
for q in "$@"; do
  GBA_BENCH_HW_QUEUES=$q timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-host --no-guides --no-fixed-band --no-end-to-end > gpurun_out/q_$q.json 2>gpurun_out/q_$q.err || exit 1
  python3 - "$q" <<'P'
import json, sys
d = json.loads(open("gpurun_out/q_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("queues", sys.argv[1], d["value"], "streamed", d["streamed"].get("gcups"), {k: (v["gcups"], (v.get("streamed") or {}).get("gcups")) for k, v in d["other_workloads"].items()})
P
done
