#!/bin/bash
# round 4: after a kernel change -- decoder + forward-golden tests, a same-box bench A/B of one knob (optional), then the rocprofv3
# kernel stats + FETCH_SIZE / WRITE_SIZE passes of the forward again (profiles/r04_forward_bf16x3_* carry the build fingerprint)
set -u
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_decoder_fused.py tests/test_decoder.py tests/test_forward.py tests/test_plan.py -q -m gpu -x 2>&1 | tail -4 | tee $OUT/r04_refresh_tests.txt
grep -q "failed\|error" $OUT/r04_refresh_tests.txt && exit 1
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
cp $OUT/r04_forward_bf16x3_pmc.json $ROOT/profiles/r04_forward_bf16x3_pmc.json
export TT_BENCH_TICK=1
timeout 600 python bench.py --no-cpu-baseline > $OUT/r04_bench_quick.json 2> $OUT/r04_bench_quick.err
cut -c1-400 $OUT/r04_bench_quick.json
grep -o '"tick_latency": {[^}]*}' $OUT/r04_bench_quick.json
grep "msda\|dec_gru\|dec_flatten\|mlp_chain" $OUT/r04_forward_bf16x3_kernel_stats.csv | cut -c1-120
