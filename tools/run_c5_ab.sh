#!/bin/bash
# usage: run_c5_ab.sh name ...   C5 (2 048 pairs of 50 kb, band 100) and its parity tests for experiment libraries ("base" = shipped)
for n in "$@"; do
  if [ "$n" = base ]; then unset GBA_LIB; else export GBA_LIB=$PWD/exp_build/libgba_$n.so; fi
  echo "== $n"
  timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q -k "c5 or wide or golden" --timeout 120 --timeout-method thread 2>&1 | tail -1
  for i in 1 2; do
    timeout -k 10 200 python3 bench.py --workload c5_50kb_wide --pairs 2048 --band 100 --steps 4 --warmup 1 --no-cpu --no-host --no-guides --no-fixed-band --no-end-to-end --no-others --no-streamed > gpurun_out/c5ab.json 2>gpurun_out/c5ab.err || exit 1
    python3 -c "
import json
d = json.loads(open('gpurun_out/c5ab.json').read().strip().splitlines()[-1])
print('gcups', d['value'], 'ms', d['ms_per_step'], 'fill_ms', d['roofline']['fill_ms'], 'tb', d['traceback_ms'], 'ok', d['config']['pairs_ok'])"
  done
done
