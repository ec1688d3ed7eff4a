#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lss.py tests/test_forward.py -m gpu -q -x -k "x3 or f32" 2>&1 | tail -4
for rr in 0 1; do TT_X3_STEM_ROWRUN=$rr TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ROWRUN=$rr', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"; done
