#!/bin/bash
# usage: run_bench_queues.sh n ...   the default bench line (no CPU baseline, no host legs) with n hardware queues (GBA_BENCH_HW_QUEUES): headline, streamed and the other configurations
cd "$GRAFT_REPO_ROOT"
for q in "$@"; do
  export GBA_BENCH_HW_QUEUES=$q
  timeout -k 10 500 python3 bench.py --no-cpu --no-host --no-end-to-end --no-fixed-band > gpurun_out/bq_$q.json 2> gpurun_out/bq_$q.err || { tail -5 gpurun_out/bq_$q.err; exit 1; }
  python3 - "$q" <<'P'
import json, sys
b = json.loads(open("gpurun_out/bq_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "queues: value", b["value"], "streamed", b["streamed"]["gcups"], "on own guides", b["guide_production"]["refine_on_these_guides"]["gcups"],
      {k: (v.get("gcups"), v.get("streamed", {}).get("gcups")) for k, v in b["other_workloads"].items()})
P
done
