#!/bin/bash
# usage: run_qv_ab.sh name1 name2 ...  tools/qv_lane_time.py on 122 880 pairs for experiment libraries ("base" = shipped)
for n in "$@"; do
  if [ "$n" = base ]; then unset GBA_LIB; else export GBA_LIB=$PWD/exp_build/libgba_$n.so; fi
  echo "== $n"
  timeout -k 10 200 python3 -u tools/qv_lane_time.py 122880 2>&1 | grep -v amdgpu.ids || exit 1
done
