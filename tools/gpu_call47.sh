#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out
cd "$ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_backward.py -m gpu -q -s -k "decoder_backward" 2>&1 | grep -v "^$" | tail -12
timeout 900 python -m pytest tests/test_train_step.py -m gpu -q -s 2>&1 | tail -15
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d "$OUT/train_trace" -o p --output-format csv -- python $ROOT/bench.py --workload train_step --steps 2 --warmup 1 --batch 8 > "$OUT/train_trace.log" 2>&1
find "$OUT/train_trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/train_step_kernel_stats.csv"
find "$OUT/train_trace" -name "*kernel_trace.csv" -delete
head -25 "$OUT/train_step_kernel_stats.csv"
