#!/bin/bash
# usage: run_others_shape.sh shape ...   the bench line's other configurations (one batch of one part each) with device-resident batches cut in the proportions `shape` (GBA_PART_SHAPE; "default": the library's own rule)
cd "$GRAFT_REPO_ROOT"
for sh in "$@"; do
  if [ "$sh" = default ]; then unset GBA_PART_SHAPE; else export GBA_PART_SHAPE=$sh; fi
  timeout -k 10 400 python3 bench.py --no-cpu --steps 2 --warmup 1 --no-host --no-guides --no-fixed-band --no-end-to-end --no-streamed > gpurun_out/others_$sh.json 2> gpurun_out/others_$sh.err || { tail -5 gpurun_out/others_$sh.err; exit 1; }
  python3 - "$sh" <<'P'
import json, sys
b = json.loads(open("gpurun_out/others_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], {k: (v.get("gcups"), v.get("ms_per_step")) for k, v in b["other_workloads"].items()})
P
done
