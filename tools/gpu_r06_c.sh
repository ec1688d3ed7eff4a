#!/bin/bash
# round 6, call C: step breakdown; fused seg head (conv tests + forward goldens); 32-wide tile timing
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_c.txt; rm -f $O
timeout 600 python -m pytest tests/test_conv.py -x -q -m gpu -k "bf16x3" 2>&1 | tail -5 | tee -a $O
timeout 900 python -m pytest tests/test_forward.py -x -q -m gpu -k "batch8 and (f32 or bf16x3) or small_matches or is_bit_repro" 2>&1 | tail -8 | tee -a $O
timeout 200 python tools/conv_microbench.py 64 224 448 64 12 3 1 x3 2>&1 | grep "^M=" | tee -a $O
timeout 200 python tools/conv_microbench.py 64 224 448 64 64 3 1 x3 2>&1 | grep "^M=" | tee -a $O
timeout 600 python tools/step_breakdown.py 8 2>&1 | grep -v amdgpu.ids | tee -a $O
