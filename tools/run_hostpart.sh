#!/bin/bash
# usage: run_hostpart.sh part_pairs ...   the bench line's host-to-host child with parts of `part_pairs` pairs (GBA_PART_PAIRS; "default": the library's own)
export GBA_NO_TORCH=1
for p in "$@"; do
  if [ "$p" = default ]; then unset GBA_PART_PAIRS; else export GBA_PART_PAIRS=$p; fi
  bash tools/run_hostchild.sh 2:16:0.4 | sed "s/^/part_pairs $p: /" || exit 1
done
