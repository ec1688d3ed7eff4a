#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for shape in "64 112 224 256 256 3" "64 28 56 512 512 3" "64 224 448 64 64 3" "64 112 224 64 256 1" "64 56 112 512 1024 1 " "64 112 224 128 256 3 2"; do
  timeout 120 python tools/wgrad_microbench.py $shape 2>&1 | grep "grad"
done
