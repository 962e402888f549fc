#!/bin/bash
# runs the non-default BASELINE configurations once each; prints one summary line per run
cd "$GRAFT_REPO_ROOT"
run() { timeout -k 10 280 python bench.py --no-cpu "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['workload'], c['pairs_per_gpu'], 'band', c['band'], 'ok', c['pairs_ok'], 'GCUPS', d['value'], 'ms', d['ms_per_step'], 'fill_ms', d['roofline']['fill_ms'], 'fill_gcups', d['roofline']['fill_gcups'], 'tb_ms', d['traceback_ms'])" || exit 1; }
run --workload c3_20kb_affine --pairs 16384 --steps 2 --warmup 1
run --workload c3_20kb_affine --pairs 53248 --steps 2 --warmup 1
run --workload c5_50kb_wide --pairs 2048 --steps 2 --warmup 1
run --workload c5_50kb_wide --pairs 512 --band 250 --steps 2 --warmup 1
run --workload c4_chr20_bestn10 --pairs 106496 --steps 2 --warmup 1
