#!/bin/bash
# round 4: kernel stats + FETCH_SIZE / WRITE_SIZE passes of the forward for the current tree (no tests), then the default bench line
set -u
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
F="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $OUT/r4f_*
TT_BENCH_DUMP=$OUT/r04_forward_bf16x3_conv_shapes.json timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4f_trace -o p -- $F > $OUT/r4f_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/r4f_fetch -o p -- $F > $OUT/r4f_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/r4f_write -o p -- $F > $OUT/r4f_write.log 2>&1
cd $ROOT
python tools/summarize_pmc.py $OUT/r4f_fetch $OUT/r4f_write > $OUT/r04_forward_bf16x3_pmc.json
cp $(find $OUT/r4f_trace -name '*kernel_stats.csv' | head -1) $OUT/r04_forward_bf16x3_kernel_stats.csv
rm -rf $OUT/r4f_fetch $OUT/r4f_write $OUT/r4f_trace
cp $OUT/r04_forward_bf16x3_pmc.json $ROOT/profiles/r04_forward_bf16x3_pmc.json
unset TT_BENCH_F32 TT_BENCH_BF16 TT_BENCH_TICK TT_BENCH_H2D TT_BENCH_VOXEL TT_BENCH_TRAIN
timeout 900 python bench.py > $OUT/r04_bench_default.json 2> $OUT/r04_bench_default.err
cut -c1-260 $OUT/r04_bench_default.json
grep -o '"tick_latency": {[^}]*}' $OUT/r04_bench_default.json
grep -o '"traffic": [0-9a-z]*' $OUT/r04_bench_default.json | head -2
