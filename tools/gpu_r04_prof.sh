#!/bin/bash
# round 4: rocprofv3 kernel stats of the forward (bench legs off, --steps 2 --warmup 1 = 4 forwards incl. the roofline pass) + conv table
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
cd /tmp && export TMPDIR=/tmp
F="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $OUT/r4p_trace
TT_BENCH_DUMP=$OUT/r04_forward_bf16x3_conv_shapes.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4p_trace -o p -- $F > $OUT/r4p_trace.log 2>&1
cp $(find $OUT/r4p_trace -name '*kernel_stats.csv' | head -1) $OUT/r04_forward_bf16x3_kernel_stats.csv
rm -rf $OUT/r4p_trace
head -40 $OUT/r04_forward_bf16x3_kernel_stats.csv | cut -c1-160
