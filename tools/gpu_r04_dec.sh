#!/bin/bash
# round 4: decoder kernels after a change: tests, phase trace of GRU / flatten, the batch-1 tick
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
O=gpurun_out/r04_dec.txt; : > $O
timeout 900 python -m pytest tests/test_decoder_fused.py tests/test_decoder.py tests/test_chain.py -q -m gpu -x 2>&1 | tail -4 | tee -a $O
python tools/dec_trace.py 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 300 python tools/tick_profile.py f32x3 10 2>&1 | grep "^tick" | tee -a $O
