#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv.py -m gpu -q > gpurun_out/r2_pytest_e_conv.log 2>&1; echo "conv rc=$?"; tail -2 gpurun_out/r2_pytest_e_conv.log
timeout 600 python tools/precision_mix_gpu.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_precision_mix.txt
for ts in 0 1; do
TT_CONV_TAIL_SPLIT=$ts TT_BENCH_F32=0 TT_BENCH_TICK=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_x3_tail$ts.json 2> gpurun_out/r2_bench_x3_tail$ts.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2_bench_x3_tail$ts.json').read().strip().splitlines()[-1])
print('tail_split=$ts x3:', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], ' bf16 leg:', d['bf16_speed_mode']['value'], d['bf16_speed_mode']['roofline']['conv_ms_per_step'])
PY
done
