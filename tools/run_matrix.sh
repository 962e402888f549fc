#!/bin/bash
# usage: run_matrix.sh "<lib>:<pairs>:<part_pairs or 0>" ...   lib = base | name in exp_build/libgba_<name>.so
for v in "$@"; do
  IFS=: read n p pp <<< "$v"
  if [ "$n" = base ]; then unset GBA_LIB; else export GBA_LIB=$PWD/exp_build/libgba_$n.so; fi
  if [ "$pp" = 0 ]; then unset GBA_PART_PAIRS; else export GBA_PART_PAIRS=$pp; fi
  timeout -k 10 300 python3 bench.py --pairs $p --steps 3 --no-cpu --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed > gpurun_out/v_${n}_${p}_${pp}.json 2>gpurun_out/v_${n}_${p}_${pp}.err || exit 1
done
