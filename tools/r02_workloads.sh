#!/bin/bash
# every BASELINE configuration through bench.py (device-resident value, guide production on the same batch)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for w in c1_ecoli_plumbing c2_10kb_linear c3_20kb_affine c4_chr20_bestn10 c5_50kb_wide; do
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu --no-host > gpurun_out/wl_$w.json 2> gpurun_out/wl_$w.err || { tail -5 gpurun_out/wl_$w.err; exit 1; }
  python - <<P
import json
d = json.loads(open("gpurun_out/wl_$w.json").read().strip().splitlines()[-1])
g = d.get("guide_production", {})
r = g.get("refine_on_these_guides", {})
print("%-18s pairs %7d  %7.1f GCUPS %8.2f ms  fill %7.2f ms  kernel %-22s | guides %7.2f ms, with guide %7d, refine on them %6.1f GCUPS, score/base %.4f (generator's guides %.4f)" % (
    "$w", d["config"]["pairs_per_gpu"], d["value"], d["ms_per_step"], d["roofline"]["fill_ms"], d["roofline"]["kernel"],
    g.get("ms_per_step", 0), g.get("pairs_with_guide", 0), r.get("gcups", 0), r.get("score_per_base", 0), r.get("score_per_base_generator_guides", 0)))
P
done
