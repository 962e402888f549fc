#!/bin/bash
# host-to-host child of bench.py with different part sizes
export GBA_NO_TORCH=1 GPU_MAX_HW_QUEUES=16
for pp in 122880 98304 81920 61440; do
  echo "== GBA_PART_PAIRS $pp"
  GBA_PART_PAIRS=$pp timeout -k 10 300 python bench.py --host-child --workload c2_10kb_linear --pairs 491520 --host-steps 6 --host-contexts 2 --host-blocks 520000000 \
    | python -c "
import json,sys
h=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('guided ms', round(h['ms_per_step'],1), 'intervals ms', round(h['intervals_ms_per_step'],1), {k: round(v['ms_per_step'],1) for k,v in h['from_sequences'].items()})
" || exit 1
done
