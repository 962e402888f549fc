#!/bin/bash
# usage: run_others_trace.sh tag ...   the bench line's other configurations under the kernel trace, once per tag: their rates and the launch timeline of the last step (the Local end rule leg)
cd "$GRAFT_REPO_ROOT"
for t in "$@"; do
  bash tools/prof_bench.sh $t --steps 2 --warmup 1 --no-host --no-guides --no-fixed-band --no-end-to-end --no-streamed > /dev/null || exit 1
  python3 - $t <<'P'
import json, sys
b = json.loads(open("gpurun_out/%s_bench.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], {k: (v.get("gcups"), v.get("traceback_ms")) for k, v in b["other_workloads"].items()})
P
  python3 tools/trace_window.py gpurun_out/prof_$t/r_kernel_trace.csv END 1 | grep -v "k_meta\|k_scan\|k_pack\|k_plan\|k_results" || exit 1
done
