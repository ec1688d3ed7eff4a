#!/bin/bash
# round 6, call E: pair-format layer timings (8 vs 4 waves on the 64-wide tile), knob-cleanup regression (conv tests)
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_e.txt; rm -f $O
timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu 2>&1 | tail -6 | tee -a $O
for shape in "64 224 448 128 64 3" "64 224 448 64 12 3" "64 112 224 64 64 3" "64 224 448 64 64 3" "64 112 224 64 256 1" "64 56 112 128 128 3" "64 28 56 512 512 3"; do
  timeout 200 python tools/conv_microbench.py $shape 1 x3 2>/dev/null | grep "^M=" | cut -c1-150 | tee -a $O
  timeout 200 python tools/conv_microbench.py $shape 1 x3p 2>/dev/null | grep "^M=" | cut -c1-150 | tee -a $O
  TT_PAIR64_WAVES=4 timeout 200 python tools/conv_microbench.py $shape 1 x3p 2>/dev/null | grep "^M=" | sed 's/^/w4 /' | cut -c1-150 | tee -a $O
done
