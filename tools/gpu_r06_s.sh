#!/bin/bash
# round 6, call S: msda_sample_proj_ln with four thread groups per row: tests, tick and batch-8 step
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_s.txt; rm -f $O
timeout 600 python -m pytest tests/test_decoder_fused.py tests/test_decoder.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O
timeout 1500 python -m pytest tests/test_forward.py tests/test_agent_tick.py tests/test_plan.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O
timeout 300 python tools/pipeline_ab.py 20 1 2>&1 | grep "in flight" | head -2 | tee -a $O
timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | head -3 | tee -a $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r6tick -o p -- python $ROOT/tools/tick_profile.py f32x3h 5 2>&1 | grep "^tick" | tee -a $O
cd $ROOT; python tools/last_tick_stats.py gpurun_out/r6tick 45 | grep "last tick\|msda\|dec_\|mlp_chain" | cut -c1-150 | tee -a $O
rm -rf gpurun_out/r6tick
