#!/bin/bash
# Kernels and copies of the host-to-host path (tools/hostpath2.py, two contexts in turn) on one time axis: the last 700 ms
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp GBA_NO_TORCH=1 GPU_MAX_HW_QUEUES=16
rm -rf gpurun_out/hp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/hp -o r -- python3 tools/hostpath2.py --batches 5 > gpurun_out/hosttrace.log 2>&1 || { tail -5 gpurun_out/hosttrace.log; exit 1; }
grep "batch\|ctxs" gpurun_out/hosttrace.log
python3 tools/trace_timeline.py gpurun_out/hp > gpurun_out/hosttrace_timeline.txt
tail -220 gpurun_out/hosttrace_timeline.txt > gpurun_out/hosttrace_tail.txt
wc -l gpurun_out/hosttrace_timeline.txt
