#!/bin/bash
# round 4: the bf16x3 split-K tile for few-row long-K layers -- conv tests, the forward goldens it touches (small F7, full-size F8
# incl. the integer parity, graph replay, sweep cache, F10 losses, F8 through the C plan), then the batch-1 tick with the path off / on
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
O=gpurun_out/r04_splitk_x3.txt; : > $O
timeout 600 python -m pytest tests/test_conv.py -q -m gpu -x -k "splitk" 2>&1 | tail -4 | tee -a $O
timeout 900 python -m pytest tests/test_forward.py tests/test_plan.py -q -m gpu -x -k "small or full_size_matches or integer_parity or graph_replay or prev_sweep or f10 or golden_f8 or reproducible" 2>&1 | tail -4 | tee -a $O
for v in 0 1 0 1; do
  echo "== TT_X3_SPLITK=$v" | tee -a $O
  TT_X3_SPLITK=$v timeout 300 python tools/tick_profile.py f32x3 10 2>&1 | grep "^tick" | tee -a $O
done
