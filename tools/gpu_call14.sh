#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
run() { timeout 200 python bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=r['static_geometry_plan']
print('$1 generic', r['avg_launch_ms'], r['frac'], 'planned', p['avg_launch_ms'], p['achieved'], p['frac'])"; }
for rf in 8 16 32; do for nt in 0 1; do TT_VP_PLAN_ROWS_IN_FLIGHT=$rf TT_VP_PLAN_NT=$nt run "rf=$rf nt=$nt"; done; done
mkdir -p gpurun_out/vp
rocprofv3 --kernel-trace --stats -d gpurun_out/vp -o vp -- python bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/vp/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
