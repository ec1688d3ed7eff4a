#!/bin/bash
# bench A/B of one environment knob:  KNOB=NAME VALS="a b" tools/gpu_r04_knob.sh   (legs off, 10 steps, each value twice)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
for rep in 1 2; do for v in $VALS; do
  env $KNOB=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$KNOB=$v', d['value'], 'frames/s', d['ms_per_step'], 'ms', 'conv', d['roofline']['conv_ms_per_step'], 'ms')
" | tee -a gpurun_out/r04_knob_$KNOB.txt
done; done
