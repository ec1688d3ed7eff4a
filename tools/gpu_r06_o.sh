#!/bin/bash
# round 6, call O: after compiling the epilogue extras out of the 16-bit tiles: conv + forward tests, mode timings
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_o.txt; rm -f $O
timeout 900 python -m pytest tests/test_conv.py tests/test_lss.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O
timeout 1800 python -m pytest tests/test_forward.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O
for d in bf16 bf16x3h bf16x3; do
  echo "TT_BENCH_DTYPE=$d" | tee -a $O
  TT_BENCH_DTYPE=$d timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | head -3 | tee -a $O
done
