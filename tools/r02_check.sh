#!/bin/bash
# GPU suite, then the default bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02_gputests.log 2>&1 || { tail -30 gpurun_out/r02_gputests.log; exit 1; }
tail -2 gpurun_out/r02_gputests.log
timeout -k 10 400 python bench.py "$@" > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err || { tail -20 gpurun_out/r02_bench.err; exit 1; }
cat gpurun_out/r02_bench.json
