#!/bin/bash
# Round-2 evidence run for the training iteration (through gpurun):
#   tools/gpu_train_r02.sh  ->  gpurun_out/train_step_b8.json, train_step_kernel_stats.csv, train_tests.log
# (tests against golden F13 / oracle autograd, the train_step bench line, rocprofv3 kernel stats of the same command)
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT" || exit 1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_conv_bwd.py tests/test_backward.py tests/test_train_step.py -m gpu -q > "$OUT/train_tests.log" 2>&1
tail -3 "$OUT/train_tests.log"
timeout 600 python bench.py --workload train_step --steps 3 --warmup 1 --batch 8 > "$OUT/train_step_b8.json" 2> "$OUT/train_step_b8.err"
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d "$OUT/train_trace" -o p --output-format csv -- \
    python "$ROOT/bench.py" --workload train_step --steps 2 --warmup 1 --batch 8 > "$OUT/train_trace.log" 2>&1
find "$OUT/train_trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/train_step_kernel_stats.csv"
find "$OUT/train_trace" -name "*kernel_trace.csv" -delete
head -c 600 "$OUT/train_step_b8.json"; echo
