#!/bin/bash
for n in "$@"; do
  if [ "$n" = base ]; then unset GBA_LIB; else export GBA_LIB=$PWD/exp_build/libgba_$n.so; fi
  echo "== $n"; timeout -k 10 400 python3 -u tools/occ_sweep.py 2>&1 | grep -v amdgpu || exit 1
done
