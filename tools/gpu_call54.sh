#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/train_trace2" -o p --output-format csv -- python $ROOT/bench.py --workload train_step --steps 2 --warmup 1 --batch 8 > "$OUT/train_trace2.log" 2>&1
find "$OUT/train_trace2" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/train_step_kernel_stats_final.csv"
find "$OUT/train_trace2" -name "*kernel_trace.csv" -delete
head -12 "$OUT/train_step_kernel_stats_final.csv" | cut -c1-160
