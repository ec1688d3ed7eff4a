#!/bin/bash
# round 6, call W: hand-pipelined h2 kernel -- bit identity against the 8-wave kernel, timing
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_w.txt; rm -f $O
timeout 600 python tools/h2_pipe_ab.py 1 2>&1 | tail -14 | tee -a $O
if [ "$QUICK" != "1" ]; then
timeout 900 python -m pytest tests/test_conv.py -x -q -m gpu -k "h2" 2>&1 | tail -3 | tee -a $O
timeout 900 python -m pytest tests/test_forward.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O
timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | head -3 | tee -a $O
fi
