#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for st in 2 3; do
  for shape in "64 224 448 64 64 3" "64 112 224 128 128 3" "64 224 448 64 64 1" "64 112 224 512 128 1" "64 112 224 256 64 1"; do
  TT_GLDS_X3_STAGES=$st timeout 120 python tools/conv_microbench.py $shape 1 x3 10 2>&1 | grep "M=" | cut -c1-70 | sed "s/^/stages=$st /"
  done
done
