#!/bin/bash
# usage: run_hostq.sh slots:q ...   the host-to-host path (tools/hostpath2.py, two contexts in turn) with GBA_SLOTS_HOST = slots at GPU_MAX_HW_QUEUES = q
export GBA_NO_TORCH=1
for c in "$@"; do
  export GBA_SLOTS_HOST=${c%%:*} GPU_MAX_HW_QUEUES=${c##*:}
  echo "== GBA_SLOTS_HOST=$GBA_SLOTS_HOST GPU_MAX_HW_QUEUES=$GPU_MAX_HW_QUEUES"
  timeout -k 10 200 python3 tools/hostpath2.py --batches 6 2>&1 | tail -1 || exit 1
done
