#!/bin/bash
# round 5, call G: 64-wide bf16x3 tile, eight waves of 32 x 64 against four of 64 x 64 (TT_GLDS_X3_64_WAVES)
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r05_g.txt; rm -f $O
for w in 8 4; do
for shape in "64 224 448 128 64 3" "64 224 448 64 64 3" "64 112 224 64 64 3" "64 112 224 256 64 1" "64 56 112 64 64 3"; do
  TT_GLDS_X3_64_WAVES=$w timeout 200 python tools/conv_microbench.py $shape 1 x3 2>/dev/null | grep "^M=" | head -1 | sed "s/^/waves=$w /" | cut -c1-90 | tee -a $O
done; done
