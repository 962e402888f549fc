#!/bin/bash
# Copies what tools/r03_final.sh left under gpurun_out/ into profiles/ under the names the closing measurements have there.
set -e
cd "$(dirname "$0")/.."
g=gpurun_out; p=profiles
cp $g/r03_final_bench.json $p/r03_final_bench.json
cp $g/r03_finalprof_kernel_stats.csv $p/r03_final_kernel_stats.csv
cp $g/r03_finalprof_bench.json $p/r03_final_kernel_stats_bench.json
{ echo "python -m pytest tests -m gpu on MI355X, the shipped build (tools/r03_final.sh):"; tail -2 $g/r03_final_gputests.log; } > $p/r03_final_gputests.txt
{ echo "rocprofv3 --pmc passes of: python3 bench.py --no-cpu --steps 1 --warmup 0 --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed (tools/r03_final.sh, tools/pmc_bench.sh); per kernel and counter: (sum over launches, launches). The lane-class kernels have seven launches: the step's five parts (half, one, one, one, half of 122 880 pairs) and the two runs of one part behind roofline.alone, so one step is 4/6 of a sum (491 520 of 737 280 pairs)."
  echo "== SQ pass"; cat $g/r03_final_sq_pmc.txt
  echo "== WRITE_SIZE pass (raw counter units, KiB)"; cat $g/r03_final_wr_pmc.txt
  echo "== FETCH_SIZE pass (raw counter units, KiB; gfx950: x 2)"; cat $g/r03_final_rd_pmc.txt; } > $p/r03_final_counters.txt
{ echo "last full step of: rocprofv3 --kernel-trace -- python3 bench.py --no-cpu --steps 2 --warmup 1 (headline leg only); start ms, end ms, duration ms, kernel, hardware queue, stream (tools/trace_last_step.py)"
  python3 tools/trace_last_step.py $g/prof_r03_finalprof/r_kernel_trace.csv 5 2; } > $p/r03_final_launch_timeline.txt
cp $g/r03_final_two_ranks.json $p/r03_final_two_ranks_gloo.json
cp $g/r03_final_two_ranks_strong.json $p/r03_final_two_ranks_gloo_strong.json
