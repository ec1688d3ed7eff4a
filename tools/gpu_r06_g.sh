#!/bin/bash
# round 6, call G: pre-split variants of the hand-pipelined kernels; 32-wide tile for the small heads; full GPU suite; step times
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_g.txt; rm -f $O
timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu 2>&1 | tail -6 | tee -a $O
for shape in "64 28 56 512 512 3" "64 56 112 128 128 3" "64 28 56 256 256 3" "64 28 56 2048 512 1"; do
  timeout 200 python tools/conv_microbench.py $shape 1 x3 2>/dev/null | grep "^M=" | cut -c1-150 | tee -a $O
  timeout 200 python tools/conv_microbench.py $shape 1 x3p 2>/dev/null | grep "^M=" | cut -c1-150 | tee -a $O
done
timeout 200 python tools/conv_microbench.py 64 28 56 512 18 3 1 x3 2>/dev/null | grep "^M=" | cut -c1-150 | tee -a $O
timeout 200 python tools/conv_microbench.py 64 224 448 64 16 1 1 x3 2>/dev/null | grep "^M=" | cut -c1-150 | tee -a $O
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_conv.py 2>&1 | tail -8 | tee -a $O
for d in bf16x3 bf16x3h; do
  echo "TT_BENCH_DTYPE=$d" | tee -a $O
  TT_BENCH_DTYPE=$d timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | head -3 | tee -a $O
done
