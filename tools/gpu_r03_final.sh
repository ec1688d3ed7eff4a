#!/bin/bash
# round 3 measurement set: default bench line, rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the forward
# (legs off, --steps 2 --warmup 1 => 4 forwards incl. the roofline pass), summarised into profiles-ready files
set -u
ROOT="$GRAFT_REPO_ROOT"
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_train_step.py -m gpu -q -x 2>&1 | tail -2
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
cd /tmp && export TMPDIR=/tmp
F="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $OUT/r3f_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3f_trace -o p -- $F > $OUT/r3f_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/r3f_fetch -o p -- $F > $OUT/r3f_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/r3f_write -o p -- $F > $OUT/r3f_write.log 2>&1
cd $ROOT
python tools/summarize_pmc.py $OUT/r3f_fetch $OUT/r3f_write > $OUT/r03_forward_bf16x3_pmc.json
cp $(find $OUT/r3f_trace -name '*kernel_stats.csv' | head -1) $OUT/r03_forward_bf16x3_kernel_stats.csv
rm -rf $OUT/r3f_fetch $OUT/r3f_write $OUT/r3f_trace      # (raw traces: tens of MB)
head -c 600 $OUT/r03_forward_bf16x3_pmc.json; echo
head -8 $OUT/r03_forward_bf16x3_kernel_stats.csv | cut -c1-200
unset TT_BENCH_F32 TT_BENCH_BF16 TT_BENCH_TICK TT_BENCH_H2D TT_BENCH_VOXEL TT_BENCH_TRAIN
# the PMC summary must sit under profiles/ for bench.py to quote it
cp $OUT/r03_forward_bf16x3_pmc.json $ROOT/profiles/r03_forward_bf16x3_pmc.json
timeout 1500 python bench.py > $OUT/r03_bench_default.json 2> $OUT/r03_bench_default.err
tail -3 $OUT/r03_bench_default.err | cut -c1-300
cut -c1-400 $OUT/r03_bench_default.json
