#!/bin/bash
# round 5, call B: agent tick tests (fixed), kernel stats of the voxel-pool op with the per-launch sort, per-phase trace of the short-K tiles
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r05_b
rm -rf $O.* $ROOT/gpurun_out/r5b_*
timeout 900 python -m pytest tests/test_agent_tick.py -q -m gpu --durations=4 2>&1 | tail -30 > $O.pytest.txt
cut -c1-300 $O.pytest.txt | tail -25
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r5b_vp -o p -- python $ROOT/bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline > $O.vp_trace.log 2>&1
cp $(find $ROOT/gpurun_out/r5b_vp -name '*kernel_stats.csv' | head -1) $O.vp_kernel_stats.csv
rm -rf $ROOT/gpurun_out/r5b_vp
head -12 $O.vp_kernel_stats.csv | cut -c1-200
cd $ROOT
timeout 600 python tools/conv_trace.py 2>&1 | tee $O.conv_trace.txt | cut -c1-400
