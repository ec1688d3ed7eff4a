#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for sp in 0 1; do
  echo "== TT_GLDS_SETPRIO=$sp"
  for shape in "64 112 224 256 256 3" "8 112 112 512 512 3" "64 56 112 256 1024 1"; do
    TT_GLDS_SETPRIO=$sp timeout 120 python tools/conv_microbench.py $shape 1 bf16 20 2>&1 | grep "M="
    TT_GLDS_SETPRIO=$sp timeout 120 python tools/conv_microbench.py $shape 1 x3 20 2>&1 | grep "M="
  done
done
