#!/bin/bash
# round 4 measurement set: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the forward (legs off, --steps 2 --warmup 1
# => 4 forwards incl. the roofline pass), the PMC summary under profiles/ (bench.py quotes it when the build fingerprint matches),
# the default bench line, and a kernel-stats profile of one training iteration.
set -u
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
F="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $OUT/r4f_*
TT_BENCH_DUMP=$OUT/r04_forward_bf16x3_conv_shapes.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4f_trace -o p -- $F > $OUT/r4f_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/r4f_fetch -o p -- $F > $OUT/r4f_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/r4f_write -o p -- $F > $OUT/r4f_write.log 2>&1
cd $ROOT
python tools/summarize_pmc.py $OUT/r4f_fetch $OUT/r4f_write > $OUT/r04_forward_bf16x3_pmc.json
cp $(find $OUT/r4f_trace -name '*kernel_stats.csv' | head -1) $OUT/r04_forward_bf16x3_kernel_stats.csv
rm -rf $OUT/r4f_fetch $OUT/r4f_write $OUT/r4f_trace
head -6 $OUT/r04_forward_bf16x3_kernel_stats.csv | cut -c1-170
cp $OUT/r04_forward_bf16x3_pmc.json $ROOT/profiles/r04_forward_bf16x3_pmc.json      # (bench.py reads it from profiles/)
# one training iteration under the profiler (bench.py --workload train_step: 1 warm-up + 1 timed + the collect() pass)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4f_train -o p -- python $ROOT/bench.py --workload train_step --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r4f_train.log 2>&1
cp $(find $OUT/r4f_train -name '*kernel_stats.csv' | head -1) $OUT/r04_train_step_b8_kernel_stats.csv
tail -1 $OUT/r4f_train.log | cut -c1-600 > $OUT/r04_train_step_b8_bench.json
rm -rf $OUT/r4f_train
head -5 $OUT/r04_train_step_b8_kernel_stats.csv | cut -c1-170
cd $ROOT
unset TT_BENCH_F32 TT_BENCH_BF16 TT_BENCH_TICK TT_BENCH_H2D TT_BENCH_VOXEL TT_BENCH_TRAIN
timeout 1500 python bench.py > $OUT/r04_bench_default.json 2> $OUT/r04_bench_default.err
tail -3 $OUT/r04_bench_default.err | cut -c1-300
cut -c1-500 $OUT/r04_bench_default.json
