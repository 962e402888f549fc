#!/bin/bash
# usage: run_hostlegs.sh slots:queues:stagger ...   the host-to-host and end-to-end legs of the bench line with GBA_SLOTS_HOST = slots, the host legs'
# GPU_MAX_HW_QUEUES = queues and the host pipeline's stagger (fraction of a batch time between two contexts' submissions)
for c in "$@"; do
  IFS=: read s q st <<< "$c"
  export GBA_SLOTS_HOST=$s GBA_BENCH_HOST_HW_QUEUES=$q BENCH_STAGGER=$st
  timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-guides --no-fixed-band --no-others --no-streamed > gpurun_out/hostlegs.json 2>gpurun_out/hostlegs.err || exit 1
  python3 - "$c" <<'P'
import json, sys
d = json.loads(open("gpurun_out/hostlegs.json").read().strip().splitlines()[-1])
h = d["host_to_host"]; e = d.get("end_to_end", {})
print(sys.argv[1], "device", d["value"], "host to host", h.get("gcups"), "intervals", (h.get("intervals_only") or {}).get("gcups"), "packed", (h.get("packed_reads") or {}).get("gcups"),
      "from sequences", {k: v.get("gcups") for k, v in (h.get("from_sequences") or {}).items() if isinstance(v, dict)},
      "end to end", {k: round(v.get("reads_per_sec", 0)) for k, v in e.items() if isinstance(v, dict)}, h.get("error"))
P
done
