#!/bin/bash
# round 2, GPU call 4: kernel trace of the bf16x3 forward with the composite decoder
ROOT="$GRAFT_REPO_ROOT"
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_r02_trace_x3" -o p --output-format csv -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/prof_r02_trace_x3.log" 2>&1
echo "trace rc=$?"
find "$OUT/prof_r02_trace_x3" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/r02_forward_x3_kernel_stats.csv"
head -40 "$OUT/r02_forward_x3_kernel_stats.csv" | cut -c1-230
# drop the bulky per-dispatch trace from what travels back (<= 64 MiB)
find "$OUT/prof_r02_trace_x3" -name "*kernel_trace.csv" -size +20M -delete
cd $ROOT
TT_BENCH_DTYPE=bf16x3 TT_DEC_FUSED=0 TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_x3_layerwise.json 2> gpurun_out/r2_bench_x3_layerwise.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench_x3_layerwise.json').read().strip().splitlines()[-1])
print('layerwise decoder:', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], json.dumps(d.get('tick_latency')))
PY
