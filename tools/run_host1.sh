#!/bin/bash
# usage: run_host1.sh shape ...   host-to-host batches through ONE context (tools/hostpath2.py --ctxs 1) with the parts in the proportions `shape` ("default": equal parts)
export GBA_NO_TORCH=1 GPU_MAX_HW_QUEUES=16
for sh in "$@"; do
  if [ "$sh" = default ]; then unset GBA_PART_SHAPE GBA_PART_SHAPE_HOST; else export GBA_PART_SHAPE=$sh GBA_PART_SHAPE_HOST=1; fi
  echo "== $sh"
  timeout -k 10 200 python3 tools/hostpath2.py --ctxs 1 --batches 4 2>&1 | tail -1 || exit 1
done
