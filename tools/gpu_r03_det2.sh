#!/bin/bash
# round 3: rewritten lift-splat strip kernel -- parity, reproducibility, per-launch time
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_voxel_pool.py tests/test_lss.py tests/test_forward.py -m gpu -x -q -k "lift_splat or lss or reproducible or small" 2>&1 | tail -8 > gpurun_out/r3_det2_pytest.txt
cat gpurun_out/r3_det2_pytest.txt
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ls -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3_det2_bench.json 2> /tmp/err.txt
f=$(find /tmp/prof_ls -name '*kernel_stats.csv' | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r3_det2_kernel_stats.csv
grep -i "lift_splat\|splitk" "$f" | cut -c1-250
tail -2 /tmp/err.txt
cut -c1-300 $GRAFT_REPO_ROOT/gpurun_out/r3_det2_bench.json
