#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TT_GLDS_X3_PINGPONG=1 timeout 600 python -m pytest tests/test_conv.py -m gpu -q -x -k "bf16x3" 2>&1 | tail -3
for pp in 0 1; do
  for shape in "64 112 224 256 256 3" "64 28 56 512 512 3" "64 56 112 256 256 3" "64 112 224 64 256 1"; do
  TT_GLDS_X3_PINGPONG=$pp timeout 120 python tools/conv_microbench.py $shape 1 x3 10 2>&1 | grep "M=" | cut -c1-70 | sed "s/^/pp=$pp /"
  done
done
