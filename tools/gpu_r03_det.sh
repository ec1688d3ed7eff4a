#!/bin/bash
# round 3: ordered split-K + atomics-free lift-splat -- parity and bit reproducibility, then lift-splat timing A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_voxel_pool.py tests/test_conv.py tests/test_lss.py tests/test_decoder.py tests/test_decoder_fused.py tests/test_plan.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_det_pytest_a.txt
timeout 900 python -m pytest tests/test_forward.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_det_pytest_b.txt
cat gpurun_out/r3_det_pytest_a.txt gpurun_out/r3_det_pytest_b.txt
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
cd /tmp && export TMPDIR=/tmp
for at in 0 1; do
  TT_LIFT_SPLAT_ATOMIC=$at timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ls$at -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3_det_bench_at$at.json 2> /tmp/err$at.txt
  f=$(find /tmp/prof_ls$at -name '*kernel_stats.csv' | head -1)
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r3_det_kernel_stats_at$at.csv
  grep -i "lift_splat\|splitk" "$f" | cut -c1-220
  tail -2 /tmp/err$at.txt
  cat $GRAFT_REPO_ROOT/gpurun_out/r3_det_bench_at$at.json | cut -c1-300
done
