#!/bin/bash
# usage: tools/prof_bench.sh <tag> [bench args...]; writes gpurun_out/<tag>_kernel_stats.csv
set -e
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o r -- python3 bench.py --no-cpu "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
cat gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
