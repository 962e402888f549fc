#!/bin/bash
# baseline on the box: GPU tests, copy rates, default bench, kernel trace, SQ counters of k_fill
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02_gputests.log 2>&1 || { tail -30 gpurun_out/r02_gputests.log; exit 1; }
tail -3 gpurun_out/r02_gputests.log
timeout -k 10 120 python tools/pcie_bw.py > gpurun_out/r02_pcie.txt 2>&1 && cat gpurun_out/r02_pcie.txt &&
timeout -k 10 300 python bench.py > gpurun_out/r02_s0_bench.json 2> gpurun_out/r02_s0_bench.err && cat gpurun_out/r02_s0_bench.json &&
bash tools/prof_bench.sh r02_s0 --steps 2 --warmup 1 &&
bash tools/pmc_bench.sh r02_s0_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SALU" --steps 1 --warmup 0
