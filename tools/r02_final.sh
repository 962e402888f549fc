#!/bin/bash
# closing measurements on the box: GPU suite, default bench line, kernel trace, counters of the fill kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02_final_gputests.log 2>&1 || { tail -30 gpurun_out/r02_final_gputests.log; exit 1; }
tail -2 gpurun_out/r02_final_gputests.log
timeout -k 10 500 python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err || { tail -20 gpurun_out/r02_final_bench.err; exit 1; }
cat gpurun_out/r02_final_bench.json
bash tools/prof_bench.sh r02_finalprof --steps 2 --warmup 1 --no-host --no-guides --no-fixed-band > /dev/null || exit 1
bash tools/pmc_bench.sh r02_final_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SALU" --steps 1 --warmup 0 --no-host --no-guides --no-fixed-band | grep -i "k_fill\|k_trace\|k_guide" || exit 1
bash tools/pmc_bench.sh r02_final_wr "WRITE_SIZE" --steps 1 --warmup 0 --no-host --no-guides --no-fixed-band | grep -i "k_fill\|k_trace" || exit 1
bash tools/pmc_bench.sh r02_final_rd "FETCH_SIZE" --steps 1 --warmup 0 --no-host --no-guides --no-fixed-band | grep -i "k_fill\|k_trace" || exit 1
timeout -k 10 300 python -u tools/kband_time.py > gpurun_out/r02_final_kband_time.log 2>&1 || exit 1
grep -v amdgpu gpurun_out/r02_final_kband_time.log
GBA_WAVE_PAIRS=1000000 timeout -k 10 300 python -u tools/qv_time.py > gpurun_out/r02_final_qv_time.log 2>&1 || exit 1
grep -v amdgpu gpurun_out/r02_final_qv_time.log
bash tools/r02_workloads.sh > gpurun_out/r02_final_workloads.log 2>&1 || { tail -5 gpurun_out/r02_final_workloads.log; exit 1; }
cat gpurun_out/r02_final_workloads.log
# two ranks on this one GPU over gloo: the sharded path bench.py --gpus N takes (common read set, merge in input order)
GBA_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --pairs 65536 > gpurun_out/r02_final_two_ranks.json 2> gpurun_out/r02_final_two_ranks.err || { tail -5 gpurun_out/r02_final_two_ranks.err; exit 1; }
cut -c1-400 gpurun_out/r02_final_two_ranks.json
