#!/bin/bash
# usage: tools/pmc_bench.sh <tag> "<counters>" [bench args...]; one counter pass, summary to gpurun_out/<tag>_pmc.txt
set -e
tag=$1; ctrs=$2; shift; shift
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf "$out"
timeout -s KILL 240 rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o r -- python3 bench.py --no-cpu "$@" > gpurun_out/${tag}_pmc_bench.json 2> gpurun_out/${tag}_pmc_bench.err
python3 tools/pmc_summary.py "$out" > gpurun_out/${tag}_pmc.txt
cat gpurun_out/${tag}_pmc.txt
