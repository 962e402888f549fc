#!/bin/bash
# usage: run_notorch.sh q ...   device-resident steps without PyTorch in the process (the system's HIP runtime) at different hardware queue counts
for q in "$@"; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  export NOTORCH_HW_QUEUES=$q
  timeout -k 10 200 python3 tools/device_notorch.py 491520 4 2>&1 | tail -1 || exit 1
  timeout -k 10 200 python3 tools/device_notorch.py 2048 3 c5_50kb_wide 100 2>&1 | tail -1 || exit 1
  timeout -k 10 200 python3 tools/device_notorch.py 16384 3 c3_20kb_affine 2>&1 | tail -1 || exit 1
  timeout -k 10 200 python3 tools/device_notorch.py 163840 3 c4_chr20_bestn10 2>&1 | tail -1 || exit 1
done
