#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv.py tests/test_lidar.py -m gpu -q > gpurun_out/r2_pytest_g.log 2>&1; echo "conv+lidar rc=$?"; grep -E "passed|failed|^E  |Error" gpurun_out/r2_pytest_g.log | cut -c1-300 | tail -12
for tp in 0 1; do
TT_SPARSE_TILE_PLAN=$tp TT_BENCH_F32=0 TT_BENCH_TICK=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0 TT_BENCH_DUMP=gpurun_out/r2_conv_shapes_x3_plan$tp.json timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_x3_plan$tp.json 2> gpurun_out/r2_bench_x3_plan$tp.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r2_bench_x3_plan$tp.json').read().strip().splitlines()[-1])
    print('plan=$tp x3:', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['algorithmic_gflop_per_step'], ' bf16 leg:', d['bf16_speed_mode']['value'], d['bf16_speed_mode']['roofline']['conv_ms_per_step'])
    rows=json.load(open('gpurun_out/r2_conv_shapes_x3_plan$tp.json'))
    for r in rows:
        if r['shape'].startswith('sparse'): print('   ', r)
except Exception as e:
    print('failed', e); print(open('gpurun_out/r2_bench_x3_plan$tp.err').read()[-1500:])
PY
done
