#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv.py -m gpu -q -x 2>&1 | tail -3
for asym in 0 1; do
  for shape in "64 112 224 256 256 3" "64 28 56 512 512 3" "64 56 112 256 256 3" "64 112 224 64 256 1"; do
  TT_GLDS_X3_ASYM=$asym timeout 120 python tools/conv_microbench.py $shape 1 x3 10 2>&1 | grep "M=" | cut -c1-70 | sed "s/^/asym=$asym /"
  done
done
