#!/bin/bash
# round 4: the last batch-1 tick under rocprofv3, eager vs issued from the C plan (per-kernel totals + span)
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r4t_*
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r4t_eager -o p -- python $ROOT/tools/tick_profile.py f32x3 5 > $OUT/r4t_eager.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r4t_plan -o p -- python $ROOT/tools/tick_profile_plan.py 5 > $OUT/r4t_plan.log 2>&1
cd $ROOT
( echo "== eager"; grep "^tick" $OUT/r4t_eager.log; python tools/last_tick_stats.py $OUT/r4t_eager 28; echo "== C plan"; grep "^tick" $OUT/r4t_plan.log; python tools/last_tick_stats.py $OUT/r4t_plan 28 ) > $OUT/r04_tick_eager_vs_plan.txt 2>&1
rm -rf $OUT/r4t_eager $OUT/r4t_plan
cut -c1-150 $OUT/r04_tick_eager_vs_plan.txt
