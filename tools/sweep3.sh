cd "$GRAFT_REPO_ROOT"
run() { env "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('%-18s pairs %7d band %3d %8.1f GCUPS  %8.2f ms  fill %8.2f ms  tb %6.2f kernel %s' % (c['workload'], c['pairs_per_gpu'], c['band'], d['value'], d['ms_per_step'], d['roofline']['fill_ms'], d['traceback_ms'], d['roofline']['kernel']))" || exit 1; }
for n in 16384 53248 98304 196608; do
run timeout -k 10 280 python bench.py --no-cpu --no-host --workload c3_20kb_affine --pairs $n --steps 2 --warmup 1
done
