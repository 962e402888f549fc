#!/bin/bash
# usage: run_sdma.sh   the host-to-host leg of the bench line with the copy engines on and off (HSA_ENABLE_SDMA)
for v in default 1 0; do
  if [ "$v" = default ]; then unset HSA_ENABLE_SDMA; else export HSA_ENABLE_SDMA=$v; fi
  timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-guides --no-fixed-band --no-end-to-end --no-others --no-streamed > gpurun_out/sdma_$v.json 2>gpurun_out/sdma_$v.err || exit 1
  python3 - "$v" <<'P'
import json, sys
d = json.loads(open("gpurun_out/sdma_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
h = d["host_to_host"]
print("HSA_ENABLE_SDMA", sys.argv[1], "device-resident", d["value"], "host to host", h.get("gcups"), h.get("ms_per_step"), "intervals only", (h.get("intervals_only") or {}).get("gcups"), h.get("error"))
P
done
