#!/bin/bash
# what the device-side CIGAR emission adds to a step: the default workload without and with it
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-cpu --no-host --steps 3 > gpurun_out/r02_cig_off.json 2> gpurun_out/r02_cig_off.err || { tail gpurun_out/r02_cig_off.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu --no-host --steps 3 --cigar 1 > gpurun_out/r02_cig_on.json 2> gpurun_out/r02_cig_on.err || { tail gpurun_out/r02_cig_on.err; exit 1; }
cat gpurun_out/r02_cig_off.json gpurun_out/r02_cig_on.json | cut -c1-400
bash tools/prof_bench.sh r02_cig --steps 2 --warmup 1 --no-host --cigar 1
