#!/bin/bash
# round 6, call Y: flatten test with 35 maps + timeline of the decoder part of a batch-1 tick
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_y.txt; rm -f $O
timeout 600 python -m pytest tests/test_decoder_fused.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r6tick -o p -- python $ROOT/tools/tick_profile.py f32x3h 5 2>&1 | grep "^tick" | tee -a $O
cd $ROOT; python tools/tick_timeline.py gpurun_out/r6tick 2 | tee -a $O
rm -rf gpurun_out/r6tick
