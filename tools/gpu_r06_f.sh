#!/bin/bash
# round 6, call F: "f32x3h" mode (bf16x3 + half-storage PAFPN): conv parity, forward goldens F7 / F8 / F14 in the new mode, step time
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_f.txt; rm -f $O
timeout 600 python -m pytest tests/test_conv.py -x -q -m gpu -k "h2 or store_half" 2>&1 | tail -5 | tee -a $O
timeout 1500 python -m pytest tests/test_forward.py -x -q -m gpu -s -k "bf16x3h" 2>&1 | grep -v amdgpu.ids | grep "rel errs\|look-module\|passed\|failed\|Error\|assert" | cut -c1-1200 | tee -a $O
for d in bf16x3h bf16x3; do
  echo "TT_BENCH_DTYPE=$d" | tee -a $O
  TT_BENCH_DTYPE=$d timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | head -3 | tee -a $O
done
