#!/bin/bash
# usage: run_variants.sh name1 name2 ...   (libs exp_build/libgba_<name>.so; "base" = shipped)
for n in "$@"; do
  if [ "$n" = base ]; then unset GBA_LIB; else export GBA_LIB=$PWD/exp_build/libgba_$n.so; fi
  timeout -k 10 200 python3 bench.py --no-cpu --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed > gpurun_out/v_$n.json 2>gpurun_out/v_$n.err || exit 1
done
