#!/bin/bash
# round 6, call T: grid2feat from the 4x4 level on as column-split per-layer launches (dec_tail_conv_kernel): test, kernel timing, tick
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_t.txt; rm -f $O
timeout 600 python -m pytest tests/test_decoder_fused.py -x -q -m gpu 2>&1 | tail -15 | tee -a $O
timeout 300 python tools/dec_microbench.py 2>&1 | grep "flatten\|gru\|bev_update" | tee -a $O
timeout 300 python tools/dec_trace.py 2>&1 | grep -A2 "flatten\|gru" | tee -a $O
if [ "$QUICK" != "1" ]; then
timeout 900 python -m pytest tests/test_decoder.py tests/test_agent_tick.py tests/test_plan.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O
timeout 300 python tools/pipeline_ab.py 20 1 2>&1 | grep "in flight" | head -2 | tee -a $O
fi
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r6tick -o p -- python $ROOT/tools/tick_profile.py f32x3h 5 2>&1 | grep "^tick" | tee -a $O
cd $ROOT; python tools/last_tick_stats.py gpurun_out/r6tick 45 | grep "last tick\|msda\|dec_\|mlp_chain" | cut -c1-150 | tee -a $O
rm -rf gpurun_out/r6tick
