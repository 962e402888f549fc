#!/bin/bash
# closing measurements on the box: GPU suite, default bench line, kernel trace, counters of the fill kernel, read harness,
# two ranks over gloo started by bench.py itself
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_final_gputests.log 2>&1 || { tail -30 gpurun_out/r03_final_gputests.log; exit 1; }
tail -2 gpurun_out/r03_final_gputests.log
timeout -k 10 600 python bench.py > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err || { tail -20 gpurun_out/r03_final_bench.err; exit 1; }
cut -c1-600 gpurun_out/r03_final_bench.json
bash tools/prof_bench.sh r03_finalprof --steps 2 --warmup 1 --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed > /dev/null || exit 1
bash tools/pmc_bench.sh r03_final_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SALU" --steps 1 --warmup 0 --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed | grep -i "k_fill\|k_trace\|k_guide" || exit 1
bash tools/pmc_bench.sh r03_final_wr "WRITE_SIZE" --steps 1 --warmup 0 --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed | grep -i "k_fill\|k_trace\|k_guide\|k_pack" || exit 1
bash tools/pmc_bench.sh r03_final_rd "FETCH_SIZE" --steps 1 --warmup 0 --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed | grep -i "k_fill\|k_trace\|k_guide\|k_pack" || exit 1
# two ranks on this one GPU over gloo, started by bench.py itself: the sharded path bench.py --gpus N takes
GBA_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --pairs 65536 --no-others --no-end-to-end --no-streamed --no-host --no-guides --no-fixed-band > gpurun_out/r03_final_two_ranks.json 2> gpurun_out/r03_final_two_ranks.err || { tail -5 gpurun_out/r03_final_two_ranks.err; exit 1; }
cut -c1-400 gpurun_out/r03_final_two_ranks.json
GBA_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --scaling strong --steps 2 --warmup 1 --pairs 131072 --no-others --no-end-to-end --no-streamed --no-host --no-guides --no-fixed-band > gpurun_out/r03_final_two_ranks_strong.json 2> gpurun_out/r03_final_two_ranks_strong.err || { tail -5 gpurun_out/r03_final_two_ranks_strong.err; exit 1; }
cut -c1-400 gpurun_out/r03_final_two_ranks_strong.json
