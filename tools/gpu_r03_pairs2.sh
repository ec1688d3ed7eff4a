#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_train_step.py -m gpu -q -s -k "f13b" 2>&1 | grep -v "^$" | tail -12 | cut -c1-600 > gpurun_out/r3_pytest_f13b.txt
cat gpurun_out/r3_pytest_f13b.txt
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
for it in 1 2 3; do
for pr in 0 1; do
  TT_X3_PAIRS=$pr timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $pr run $it', b['value'], 'frames/s', b['ms_per_step'], 'ms', 'conv ms', b['roofline'].get('kernel_ms'), b['roofline']['frac'])"
done
done
