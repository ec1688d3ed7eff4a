#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT || exit 1
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_voxel_pool.py -m gpu -q 2>&1 | tail -3
timeout 200 python bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee $OUT/r02_voxel_pool_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=r['static_geometry_plan']
print('generic', r['avg_launch_ms'], r['frac'], 'planned', p['avg_launch_ms'], p['achieved'], p['frac'], 'value', d['value'])"
cd /tmp
for mode in f32x3; do
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tick_$mode -o p --output-format csv -- python $ROOT/tools/tick_profile.py $mode 5 > $OUT/tick_$mode.log 2>&1
tail -1 $OUT/tick_$mode.log
find $OUT/tick_$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r02_tick_${mode}_kernel_stats.csv
find $OUT/tick_$mode -name "*kernel_trace.csv" -delete
done
