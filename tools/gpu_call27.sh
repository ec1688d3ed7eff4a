#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv.py tests/test_lidar.py -m gpu -q -x 2>&1 | tail -3
for pipe in 0 1; do
  for shape in "64 112 224 256 256 3" "64 28 56 512 512 3" "64 224 448 64 64 3" "64 112 224 128 128 3" "64 112 224 64 256 1"; do
  TT_GLDS_X3_PIPE=$pipe timeout 120 python tools/conv_microbench.py $shape 1 x3 10 2>&1 | grep "M=" | cut -c1-70 | sed "s/^/pipe=$pipe /"
  done
done
for pipe in 0 1; do TT_GLDS_X3_PIPE=$pipe TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench PIPE=$pipe', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['dominant_kernel']['executed_mfma_frac'])"; done
