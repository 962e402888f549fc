#!/bin/bash
# usage: run_hostchild.sh slots:queues:stagger ...   the bench line's host-to-host child on its own (no parent process holding hardware queues)
export GBA_NO_TORCH=1
for c in "$@"; do
  IFS=: read s q st <<< "$c"
  export GBA_SLOTS_HOST=$s GPU_MAX_HW_QUEUES=$q BENCH_STAGGER=$st
  timeout -k 10 400 python3 bench.py --host-child --workload c2_10kb_linear --pairs 491520 --host-steps 6 --host-contexts 2 --host-blocks 520000000 > gpurun_out/hostchild.json 2>gpurun_out/hostchild.err || { tail -3 gpurun_out/hostchild.err; exit 1; }
  python3 - "$c" <<'P'
import json, sys
h = json.loads(open("gpurun_out/hostchild.json").read().strip().splitlines()[-1])
g = lambda ms: round(h["cells"] / (ms / 1e3) / 1e9, 1)
print(sys.argv[1], "host to host", g(h["ms_per_step"]), "intervals", g(h["intervals_ms_per_step"]), "packed", g(h["packed_ms_per_step"]), "from sequences", {k: round(v["cells"] / (v["ms_per_step"] / 1e3) / 1e9, 1) for k, v in h["from_sequences"].items()})
P
done
