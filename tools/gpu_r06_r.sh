#!/bin/bash
# round 6, call R: per-kernel totals of one batch-1 tick (headline mode)
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_tick_last_tick.txt; rm -f $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r6tick -o p -- python $ROOT/tools/tick_profile.py f32x3h 5 2>&1 | grep "^tick" | tee -a $O
cd $ROOT; python tools/last_tick_stats.py gpurun_out/r6tick 45 | cut -c1-170 | tee -a $O
rm -rf gpurun_out/r6tick
