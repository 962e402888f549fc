#!/bin/bash
# usage: [PAIRS=n] run_host2.sh shape ...   host-to-host batches through two contexts in turn (tools/hostpath2.py) with the parts in the proportions `shape` ("default": equal parts of about 122 880)
export GBA_NO_TORCH=1 GPU_MAX_HW_QUEUES=16
for sh in "$@"; do
  if [ "$sh" = default ]; then unset GBA_PART_SHAPE GBA_PART_SHAPE_HOST; else export GBA_PART_SHAPE=$sh GBA_PART_SHAPE_HOST=1; fi
  echo "== $sh"
  timeout -k 10 200 python3 tools/hostpath2.py --ctxs 2 --batches 6 --pairs ${PAIRS:-491520} 2>&1 | tail -1 || exit 1
done
