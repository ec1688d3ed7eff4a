#!/bin/bash
# round 4 quick check: forward parity tests + conv tests, then a short bench (legs off) with and without the pipelined tile
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_conv.py tests/test_forward.py tests/test_lss.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r04_quick_pytest.txt
cat gpurun_out/r04_quick_pytest.txt
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
for pipe in 0 2 0 2; do
  TT_X3_PIPE=$pipe timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pipe=$pipe', d['value'], 'frames/s', d['ms_per_step'], 'ms', 'conv', d['roofline']['conv_ms_per_step'], 'ms', d['roofline']['achieved'], 'TF/s', d['roofline']['dominant_kernel']['kernel'], d['roofline']['dominant_kernel']['ms_per_step'])
" | tee -a gpurun_out/r04_quick_bench.txt
done
