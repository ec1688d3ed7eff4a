#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT || exit 1
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_voxel_pool.py tests/test_lss.py -m gpu -q 2>&1 | tail -3
run() { timeout 200 python bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee $OUT/r02_voxel_pool_$1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=r['static_geometry_plan']
print('$1 generic', r['avg_launch_ms'], r['frac'], 'planned', p['avg_launch_ms'], p['achieved'], p['frac'], 'value', d['value'])"; }
run nt1
TT_VP_NT=0 run nt0
cd /tmp
B="python $ROOT/bench.py --workload voxel_pool --steps 10 --warmup 3 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/vp_trace -o p --output-format csv -- $B > $OUT/vp_trace.log 2>&1
find $OUT/vp_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r02_voxel_pool_kernel_stats.csv
find $OUT/vp_trace -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/vp_pmc_$c -o p --output-format csv -- $B > $OUT/vp_pmc_$c.log 2>&1
done
python $ROOT/tools/summarize_pmc.py $OUT/vp_pmc_FETCH_SIZE $OUT/vp_pmc_WRITE_SIZE > $OUT/r02_voxel_pool_pmc.json
head -8 $OUT/r02_voxel_pool_kernel_stats.csv | cut -c1-150
head -c 1500 $OUT/r02_voxel_pool_pmc.json
