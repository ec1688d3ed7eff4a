#!/bin/bash
# round 6, call D: pair-format activations -- parity (bit identity), layer timings, forward goldens, step time
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_d.txt; rm -f $O
timeout 600 python -m pytest tests/test_conv.py -x -q -m gpu -k "pair or bf16x3" 2>&1 | tail -8 | tee -a $O
timeout 900 python -m pytest tests/test_forward.py tests/test_lss.py -x -q -m gpu -k "not f16 and not bf16-" 2>&1 | tail -8 | tee -a $O
for p in 1 0; do
  echo "TT_X3_PAIR=$p" | tee -a $O
  TT_X3_PAIR=$p timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | head -3 | tee -a $O
done
