#!/bin/bash
# round 2, GPU call 5: composite decoder after the prefetch-ring fix: unit tests, trace, tick latency
ROOT="$GRAFT_REPO_ROOT"
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd $ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_chain.py tests/test_decoder_fused.py tests/test_decoder.py -m gpu -q > gpurun_out/r2_pytest_d.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r2_pytest_d.log
cd /tmp
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_r02_trace_x3b" -o p --output-format csv -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/prof_r02_trace_x3b.log" 2>&1
echo "trace rc=$?"
find "$OUT/prof_r02_trace_x3b" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/r02_forward_x3b_kernel_stats.csv"
grep -E "mlp_chain|dec_|msda|look_|sca_|concat" "$OUT/r02_forward_x3b_kernel_stats.csv" | cut -c1-200
find "$OUT/prof_r02_trace_x3b" -name "*kernel_trace.csv" -size +20M -delete
cd $ROOT
TT_BENCH_TICK=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_x3_c.json 2> gpurun_out/r2_bench_x3_c.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench_x3_c.json').read().strip().splitlines()[-1])
print('x3 composite:', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], json.dumps(d.get('tick_latency')))
PY
