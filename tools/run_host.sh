#!/bin/bash
# usage: run_host.sh <lib> ...   headline + host path per library (base = shipped)
for n in "$@"; do
  if [ "${n%%:*}" = base ]; then unset GBA_LIB; else export GBA_LIB=$PWD/exp_build/libgba_${n%%:*}.so; fi
  timeout -k 10 400 python3 bench.py --no-cpu --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed > gpurun_out/h_$n.json 2>gpurun_out/h_$n.err || exit 1
done
