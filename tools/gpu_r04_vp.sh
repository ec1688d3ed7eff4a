#!/bin/bash
# round 4: B1 voxel-pool operator at the thinktwice.py size (8 samples per launch): generic two-phase kernel and the static-geometry plan
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2 3; do
python bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('generic', r['avg_launch_ms'], 'ms frac', r['frac'], '| planned', r['static_geometry_plan']['avg_launch_ms'], 'ms frac', r['static_geometry_plan']['frac'])
" | tee -a gpurun_out/r04_vp.txt
done
timeout 600 python -m pytest tests/test_voxel_pool.py -q -m gpu -x 2>&1 | tail -2 | tee -a gpurun_out/r04_vp.txt
