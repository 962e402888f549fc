#!/bin/bash
# guide production at the default workload: bench line with the guide_production object, kernel times
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-cpu --no-host --steps 2 "$@" > gpurun_out/r02_sdp_bench.json 2> gpurun_out/r02_sdp_bench.err || { tail gpurun_out/r02_sdp_bench.err; exit 1; }
python - <<'P'
import json
d = json.loads(open("gpurun_out/r02_sdp_bench.json").read().strip().splitlines()[-1])
print(d["value"], json.dumps(d.get("guide_production"), indent=1))
P
bash tools/prof_bench.sh r02_sdp --steps 1 --warmup 0 --no-host "$@" | grep -i "sdp\|pack" | cut -c1-220
