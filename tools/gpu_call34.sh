#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rf in 8 16 32; do TT_VP_ROWS_IN_FLIGHT=$rf timeout 200 python bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=r['static_geometry_plan']
print('rf=$rf generic', r['avg_launch_ms'], r['frac'], 'planned', p['avg_launch_ms'], p['frac'])"; done
