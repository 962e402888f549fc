#!/bin/bash
# SQ counters of the fixed-band kernels on tools/kband_time.py; summary to gpurun_out/kband_pmc.txt
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
out=gpurun_out/pmc_kband
rm -rf "$out"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d "$out" -o r -- python3 tools/kband_time.py > gpurun_out/kband_pmc.log 2>&1 || exit 1
python3 tools/pmc_summary.py "$out" > gpurun_out/kband_pmc.txt
cat gpurun_out/kband_pmc.txt
rm -rf "$out"
