#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for mode in f32x3; do
timeout 300 rocprofv3 --kernel-trace -d $OUT/tick2_$mode -o p --output-format csv -- python $ROOT/tools/tick_profile.py $mode 5 > $OUT/tick2_$mode.log 2>&1
tail -1 $OUT/tick2_$mode.log
python $ROOT/tools/last_tick_stats.py $OUT/tick2_$mode 60 > $OUT/r02_tick_${mode}_last_tick.txt
find $OUT/tick2_$mode -name "*kernel_trace.csv" -delete
done
