#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_voxel_pool.py tests/test_agent_tick.py tests/test_forward.py -m gpu -q -k "voxel or agent or lidar or mmcv" 2>&1 | tail -3
for rf in 4 8 16; do
  TT_VP_ROWS_IN_FLIGHT=$rf timeout 200 python bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('RF=$rf', 'avg_launch_ms', r['avg_launch_ms'], 'compulsory GB/s', r['achieved'], 'frac', r['frac'])"
done
