#!/bin/bash
# round 6, call H: the failing training golden in detail; sample-first decoder (unit test, forward goldens, step time both ways)
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_h.txt; rm -f $O
timeout 900 python -m pytest tests/test_train_step.py -x -q -m gpu -k "f13b" 2>&1 | grep -v amdgpu.ids | tail -40 | cut -c1-600 | tee -a $O
timeout 600 python -m pytest tests/test_decoder_fused.py tests/test_decoder.py -x -q -m gpu 2>&1 | tail -6 | tee -a $O
timeout 1500 python -m pytest tests/test_forward.py tests/test_agent_tick.py tests/test_plan.py -x -q -m gpu 2>&1 | tail -6 | tee -a $O
for sf in 1 0; do
  echo "TT_DEC_SAMPLE_FIRST=$sf" | tee -a $O
  TT_DEC_SAMPLE_FIRST=$sf timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | head -3 | tee -a $O
done
timeout 200 python tools/conv_microbench.py 64 56 112 512 512 1 1 x3 2>/dev/null | grep "^M=\|copy" | cut -c1-150 | tee -a $O
