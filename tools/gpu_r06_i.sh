#!/bin/bash
# round 6, call I: sample-first decoder (tests, forward goldens, step time both ways), training goldens
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_i.txt; rm -f $O
timeout 600 python -m pytest tests/test_decoder_fused.py tests/test_decoder.py -x -q -m gpu 2>&1 | tail -6 | tee -a $O
timeout 1500 python -m pytest tests/test_forward.py tests/test_agent_tick.py tests/test_plan.py -x -q -m gpu 2>&1 | tail -6 | tee -a $O
timeout 1500 python -m pytest tests/test_train_step.py -x -q -m gpu 2>&1 | tail -6 | tee -a $O
for sf in 1 0; do
  echo "TT_DEC_SAMPLE_FIRST=$sf" | tee -a $O
  TT_DEC_SAMPLE_FIRST=$sf timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | head -3 | tee -a $O
done
TT_BENCH_DTYPE=bf16x3 timeout 300 python tools/pipeline_ab.py 20 1 2>&1 | grep "in flight" | head -2 | tee -a $O
TT_DEC_SAMPLE_FIRST=0 timeout 300 python tools/pipeline_ab.py 20 1 2>&1 | grep "in flight" | head -1 | tee -a $O
