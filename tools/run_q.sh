#!/bin/bash
# usage: run_q.sh "<lib>:<hw queues>[:<tag>]" ...
for v in "$@"; do
  IFS=: read n q t <<< "$v"
  if [ "$n" = base ]; then unset GBA_LIB; else export GBA_LIB=$PWD/exp_build/libgba_$n.so; fi
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 bench.py --no-cpu --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed > gpurun_out/v_${n}_q$q$t.json 2>gpurun_out/v_${n}_q$q$t.err || exit 1
done
