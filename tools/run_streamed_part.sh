#!/bin/bash
# usage: run_streamed_part.sh part_pairs ...   headline and two-context streamed rate of the bench line with parts of `part_pairs` pairs (GBA_PART_PAIRS; "default": the library's own)
cd "$GRAFT_REPO_ROOT"
for p in "$@"; do
  if [ "$p" = default ]; then unset GBA_PART_PAIRS; else export GBA_PART_PAIRS=$p; fi
  timeout -k 10 300 python3 bench.py --no-cpu --no-host --no-end-to-end --no-fixed-band --no-others --no-guides > gpurun_out/sp_$p.json 2> gpurun_out/sp_$p.err || { tail -5 gpurun_out/sp_$p.err; exit 1; }
  python3 - "$p" <<'P'
import json, sys
b = json.loads(open("gpurun_out/sp_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("part_pairs", sys.argv[1], "value", b["value"], "streamed", b["streamed"]["gcups"])
P
done
