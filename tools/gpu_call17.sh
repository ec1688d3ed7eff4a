#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv.py -m gpu -q -x 2>&1 | tail -3
for t in 1 0 64 128; do echo "X3_TILE=$t"; TT_GLDS_X3_TILE=$t timeout 300 python tools/tick_profile.py f32x3 10 2>&1 | tail -1; done
for t in 1 0; do TT_GLDS_X3_TILE=$t TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench X3_TILE=$t', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"; done
