#!/bin/bash
# lane-per-pair against wave-per-pair kernels at several batch sizes
cd "$GRAFT_REPO_ROOT"
run() { env "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('%-18s pairs %7d  %8.1f GCUPS  %8.2f ms  fill %8.2f ms  kernel %s' % (c['workload'], c['pairs_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['fill_ms'], d['roofline']['kernel']))" || exit 1; }
for n in 2048 8192 16384 32768; do for m in 0 4000000000; do
  run GBA_WAVE_PAIRS=$m GBA_WAVE_PAIRS_AFFINE=$m timeout -k 10 280 python bench.py --no-cpu --no-host --workload c2_10kb_linear --pairs $n --steps 3 --warmup 1
done; done
for n in 4096 16384 32768 53248; do for m in 0 4000000000; do
  run GBA_WAVE_PAIRS=$m GBA_WAVE_PAIRS_AFFINE=$m timeout -k 10 280 python bench.py --no-cpu --no-host --workload c3_20kb_affine --pairs $n --steps 3 --warmup 1
done; done
run timeout -k 10 280 python bench.py --no-cpu --no-host --workload c5_50kb_wide --pairs 2048 --steps 2 --warmup 1
run timeout -k 10 280 python bench.py --no-cpu --no-host --workload c5_50kb_wide --pairs 512 --band 250 --steps 2 --warmup 1
run timeout -k 10 280 python bench.py --no-cpu --no-host --workload c4_chr20_bestn10 --steps 2 --warmup 1
