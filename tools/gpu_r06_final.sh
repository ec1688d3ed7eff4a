#!/bin/bash
# round 6 measurement set (run as a whole at a checkpoint and again at the end; STAGE=tests|prof|bench picks one part):
#   tests : the whole -m gpu suite + smoke()
#   prof  : rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the forward (legs off, --steps 2 --warmup 1 => 4 forwards incl.
#           the roofline pass) and of the voxel-pool op; summaries under profiles/r06_* (bench.py quotes the forward PMC when the build
#           fingerprint matches)
#   bench : the default bench line
set -u
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
STAGE=${STAGE:-all}
cd $ROOT
if [ "$STAGE" = all ] || [ "$STAGE" = tests ]; then
  timeout 2400 python -m pytest tests/ -q -m gpu --maxfail=10 --durations=8 2>&1 | tail -30 > $OUT/r06_pytest_gpu.txt
  cut -c1-300 $OUT/r06_pytest_gpu.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $OUT/r06_pytest_gpu.txt
fi
if [ "$STAGE" = all ] || [ "$STAGE" = prof ]; then
  cd /tmp && export TMPDIR=/tmp
  export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_X3=0 TT_BENCH_SERIAL=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_RAW=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
  F="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
  V="python $ROOT/bench.py --workload voxel_pool --steps 10 --warmup 3 --no-cpu-baseline"
  rm -rf $OUT/r6f_*
  TT_BENCH_DUMP=$OUT/r06_forward_bf16x3h_conv_shapes.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6f_trace -o p -- $F > $OUT/r6f_trace.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/r6f_fetch -o p -- $F > $OUT/r6f_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/r6f_write -o p -- $F > $OUT/r6f_write.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6f_vtrace -o p -- $V > $OUT/r6f_vtrace.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/r6f_vfetch -o p -- $V > $OUT/r6f_vfetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/r6f_vwrite -o p -- $V > $OUT/r6f_vwrite.log 2>&1
  cd $ROOT
  python tools/summarize_pmc.py $OUT/r6f_fetch $OUT/r6f_write > $OUT/r06_forward_bf16x3h_pmc.json
  python tools/summarize_pmc.py $OUT/r6f_vfetch $OUT/r6f_vwrite > $OUT/r06_voxel_pool_pmc.json
  cp $(find $OUT/r6f_trace -name '*kernel_stats.csv' | head -1) $OUT/r06_forward_bf16x3h_kernel_stats.csv
  cp $(find $OUT/r6f_vtrace -name '*kernel_stats.csv' | head -1) $OUT/r06_voxel_pool_kernel_stats.csv
  tail -1 $OUT/r6f_vtrace.log | cut -c1-1500 > $OUT/r06_voxel_pool_bench.json
  rm -rf $OUT/r6f_fetch $OUT/r6f_write $OUT/r6f_trace $OUT/r6f_vtrace $OUT/r6f_vfetch $OUT/r6f_vwrite
  head -8 $OUT/r06_forward_bf16x3h_kernel_stats.csv | cut -c1-170
  grep -E "vp_|voxel" $OUT/r06_voxel_pool_kernel_stats.csv | cut -c1-150
  cp $OUT/r06_forward_bf16x3h_pmc.json $ROOT/profiles/r06_forward_bf16x3h_pmc.json      # (bench.py reads it from profiles/)
  unset TT_BENCH_F32 TT_BENCH_BF16 TT_BENCH_X3 TT_BENCH_SERIAL TT_BENCH_TICK TT_BENCH_H2D TT_BENCH_RAW TT_BENCH_VOXEL TT_BENCH_TRAIN
fi
if [ "$STAGE" = all ] || [ "$STAGE" = bench ]; then
  cd $ROOT
  timeout 1500 python bench.py > $OUT/r06_bench_default.json 2> $OUT/r06_bench_default.err
  tail -3 $OUT/r06_bench_default.err | cut -c1-300
  cut -c1-600 $OUT/r06_bench_default.json
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_default.json"))
e = d.get("extra", {})
for k in ("one_batch_at_a_time", "bf16x3_mode", "tick_latency", "raw_inclusive", "h2d_inclusive", "lift_splat", "cpu_baseline"):
    v = e.get(k, d.get(k))
    print(k, json.dumps(v)[:600])
v = e.get("voxel_pool_op", {})
print("voxel_pool_op", {k: v.get(k) for k in ("avg_launch_ms", "frac", "achieved", "static_geometry_plan")})
print("train_step", json.dumps(e.get("train_step"))[:400])
PY
fi
