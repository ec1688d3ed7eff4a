#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_voxel_pool.py tests/test_agent_tick.py -m gpu -q 2>&1 | tail -6
timeout 200 python bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('generic', r['avg_launch_ms'], r['achieved'], r['frac']); print('planned', json.dumps(r['static_geometry_plan'])[:400])"
