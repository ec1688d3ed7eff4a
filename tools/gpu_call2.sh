#!/bin/bash
# round 2, GPU call 2: bf16x3 precision mode (conv unit tests, model-level parity, bench), f16 re-check
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv.py -m gpu -q -k "x3 or f32x3 or float16" > gpurun_out/r2_pytest_b_conv.log 2>&1; echo "conv rc=$?"
tail -15 gpurun_out/r2_pytest_b_conv.log
timeout 900 python -m pytest tests/test_lss.py tests/test_forward.py tests/test_optim.py tests/test_decoder.py -m gpu -q -s > gpurun_out/r2_pytest_b_model.log 2>&1; echo "model rc=$?"
grep -E "passed|failed|rel errs|torch.float|f32x3|Error" gpurun_out/r2_pytest_b_model.log | cut -c1-900 | tail -30
TT_BENCH_DUMP=gpurun_out/r2_conv_shapes_x3_a.json timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2_bench_x3_a.json 2> gpurun_out/r2_bench_x3_a.err; echo "bench x3 rc=$?"
tail -3 gpurun_out/r2_bench_x3_a.err
for v in 6 7; do
  for shape in "64 112 224 256 256 3" "8 112 112 512 512 3"; do
    TT_GLDS_VARIANT=$v timeout 120 python tools/conv_microbench.py $shape 1 bf16 20 2>&1 | grep "M="
  done
done > gpurun_out/r2_microbench_v7.txt 2>&1
cat gpurun_out/r2_microbench_v7.txt
head -c 600 gpurun_out/r2_bench_x3_a.json
