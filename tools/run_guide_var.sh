#!/bin/bash
# usage: run_guide_var.sh lib ...   k_guide's launch duration alone (a batch of one part, 81 920 pairs of 10 kb) for experiment builds tools/exp/libgba_<lib>.so
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
A="--pairs 81920 --steps 3 --warmup 1 --no-host --no-guides --no-fixed-band --no-others --no-end-to-end --no-streamed"
for lib in "$@"; do
  export GBA_LIB=$PWD/tools/exp/libgba_$lib.so
  bash tools/prof_bench.sh gv_$lib $A | grep -i "k_guide" | sed 's/"[^"]*"/k_guide/' | sed "s/^/$lib /" || exit 1
done
