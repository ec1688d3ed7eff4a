#!/bin/bash
# round 5, call C: epilogue residual prefetch (tests + A/B numbers + tile trace), lane-mask counting sort (tests + op numbers + kernel stats),
# stage-wise agent tick tests
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r05_c
rm -rf $O.* $ROOT/gpurun_out/r5c_*
timeout 1500 python -m pytest tests/test_agent_tick.py tests/test_voxel_pool.py tests/test_conv.py -q -m gpu --maxfail=8 --durations=4 2>&1 | tail -40 > $O.pytest.txt
cut -c1-300 $O.pytest.txt | tail -30
for i in 1 2 3; do
timeout 300 python bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('generic (per-launch counting sort)', r['avg_launch_ms'], 'ms frac', r['frac'], '| planned', r['static_geometry_plan']['avg_launch_ms'], 'ms frac', r['static_geometry_plan']['frac'])
" | tee -a $O.vp.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r5c_vp -o p -- python $ROOT/bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline > $O.vp_trace.log 2>&1
cp $(find $ROOT/gpurun_out/r5c_vp -name '*kernel_stats.csv' | head -1) $O.vp_kernel_stats.csv
rm -rf $ROOT/gpurun_out/r5c_vp
grep -E "vp_|voxel" $O.vp_kernel_stats.csv | cut -c1-160
cd $ROOT
timeout 600 python tools/shortk_ab.py base 2>&1 | tee $O.shortk.txt | cut -c1-200
timeout 600 python tools/conv_trace.py 2>&1 | grep -v "workgroups inside" | tee $O.conv_trace.txt | cut -c1-300
