#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r2_pytest_f_all.log 2>&1; echo "all tests rc=$?"; tail -4 gpurun_out/r2_pytest_f_all.log
TT_BENCH_F32=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_x3_d.json 2> gpurun_out/r2_bench_x3_d.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_bench_x3_d.json').read().strip().splitlines()[-1])
print('x3:', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], json.dumps(d.get('tick_latency')), ' bf16 leg:', d['bf16_speed_mode']['value'], d['bf16_speed_mode']['ms_per_step'])
PY
