#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_cigar.py tests/test_gpu_intervals.py tests/test_zz_no_alloc.py -x -q --timeout 200 --timeout-method thread > gpurun_out/tbw_tests.log 2>&1 || { tail -30 gpurun_out/tbw_tests.log; exit 1; }
tail -2 gpurun_out/tbw_tests.log
timeout -k 10 150 python -u tests/fuzz_gpu.py --seconds 90 --seed 31337 > gpurun_out/tbw_fuzz.log 2>&1 || { tail -5 gpurun_out/tbw_fuzz.log; exit 1; }
tail -2 gpurun_out/tbw_fuzz.log
for b in 100 250; do
  n=2048; [ $b = 250 ] && n=512
  timeout -k 10 200 python bench.py --workload c5_50kb_wide --band $b --pairs $n --steps 3 --warmup 1 --no-cpu --no-host --no-guides --no-fixed-band > gpurun_out/tbw_$b.json 2> gpurun_out/tbw_$b.err || exit 1
  echo "band $b: $(python -c "import json; d=json.load(open('gpurun_out/tbw_$b.json')); print(d['value'], 'GCUPS', d['ms_per_step'], 'ms, fill', d['roofline']['fill_ms'], 'tb', d['traceback_ms'], d['roofline']['kernel'])")"
done
