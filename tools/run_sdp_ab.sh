#!/bin/bash
# usage: run_sdp_ab.sh name1 name2 ...  phase times of gba_sdp_device for experiment libraries ("base" = shipped)
for n in "$@"; do
  if [ "$n" = base ]; then unset GBA_LIB; else export GBA_LIB=$PWD/exp_build/libgba_$n.so; fi
  echo "== $n"
  timeout -k 10 200 python3 tools/sdp_time.py 32768 || exit 1
done
