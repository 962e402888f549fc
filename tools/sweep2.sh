cd "$GRAFT_REPO_ROOT"
run() { env "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('%-18s pairs %7d band %3d %8.1f GCUPS  %8.2f ms  fill %8.2f ms  tb %6.2f kernel %s' % (c['workload'], c['pairs_per_gpu'], c['band'], d['value'], d['ms_per_step'], d['roofline']['fill_ms'], d['traceback_ms'], d['roofline']['kernel']))" || exit 1; }
run timeout -k 10 280 python bench.py --no-cpu --no-host --workload c5_50kb_wide --pairs 2048 --steps 2 --warmup 1
run timeout -k 10 280 python bench.py --no-cpu --no-host --workload c5_50kb_wide --pairs 512 --band 250 --steps 2 --warmup 1
run timeout -k 10 280 python bench.py --no-cpu --no-host --workload c3_20kb_affine --steps 3 --warmup 1
run timeout -k 10 280 python bench.py --no-cpu --no-host --workload c3_20kb_affine --pairs 98304 --steps 2 --warmup 1
run GBA_WAVE_PAIRS_AFFINE=0 timeout -k 10 280 python bench.py --no-cpu --no-host --workload c3_20kb_affine --pairs 98304 --steps 2 --warmup 1
run timeout -k 10 280 python bench.py --no-cpu --no-host --workload c1_ecoli_plumbing --pairs 4096 --steps 3 --warmup 1
run GBA_WAVE_PAIRS=0 timeout -k 10 280 python bench.py --no-cpu --no-host --workload c1_ecoli_plumbing --pairs 4096 --steps 3 --warmup 1
