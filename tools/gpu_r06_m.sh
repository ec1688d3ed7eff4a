#!/bin/bash
# round 6, call M: f32x3h with f32 sum chains + half conv-input copies: conv parity, forward goldens in that mode, step time
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_m.txt; rm -f $O
timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu 2>&1 | tail -4 | tee -a $O
timeout 1500 python -m pytest tests/test_forward.py -x -q -m gpu -s -k "bf16x3h" 2>&1 | grep -v amdgpu.ids | grep "rel errs\|look-module\|passed\|failed\|Error\|assert" | cut -c1-1400 | tee -a $O
for d in bf16x3h bf16x3; do
  echo "TT_BENCH_DTYPE=$d" | tee -a $O
  TT_BENCH_DTYPE=$d timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | head -3 | tee -a $O
done
