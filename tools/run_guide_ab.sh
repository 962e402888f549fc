#!/bin/bash
# usage: run_guide_ab.sh   the default step with the library in the tree and with tools/exp/libgba_base.so (GBA_LIB), in turn: GCUPS and k_guide's launch durations
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed"
for lib in new base new base; do
  if [ $lib = base ]; then export GBA_LIB=$PWD/tools/exp/libgba_base.so; else unset GBA_LIB; fi
  timeout -k 10 200 python3 bench.py --no-cpu $A > gpurun_out/gab_$lib.json 2> gpurun_out/gab_$lib.err || { tail -5 gpurun_out/gab_$lib.err; exit 1; }
  python3 -c "
import json,sys
b=json.loads(open('gpurun_out/gab_$lib.json').read().strip().splitlines()[-1])
print('$lib', b['value'], b['ms_per_step'], b['roofline']['fill_ms'])"
done
for lib in new base; do
  if [ $lib = base ]; then export GBA_LIB=$PWD/tools/exp/libgba_base.so; else unset GBA_LIB; fi
  bash tools/prof_bench.sh gab_$lib $A | grep -i "k_guide\|k_fill<\|k_traceback<" | cut -d, -f1-6 | sed "s/^/$lib /" || exit 1
done
