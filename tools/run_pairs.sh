#!/bin/bash
# usage: run_pairs.sh "<pairs>:<part_pairs or 0>" ...
for v in "$@"; do
  p=${v%%:*}; pp=${v#*:}
  if [ "$pp" = 0 ]; then unset GBA_PART_PAIRS; else export GBA_PART_PAIRS=$pp; fi
  timeout -k 10 300 python3 bench.py --pairs $p --steps 3 --no-cpu --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed > gpurun_out/v_p${p}_${pp}.json 2>gpurun_out/v_p${p}_${pp}.err || exit 1
done
