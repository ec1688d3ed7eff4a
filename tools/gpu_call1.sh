#!/bin/bash
# round 2, GPU call 1: baseline tests (incl. new f16 / B=8 tests), f16 + bf16 bench lines, never-run variants
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2_pytest_a.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r2_pytest_a.log
TT_BENCH_DUMP=gpurun_out/r2_conv_shapes_f16_a.json timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2_bench_f16_a.json 2> gpurun_out/r2_bench_f16_a.err; echo "bench f16 rc=$?"
TT_BENCH_DTYPE=bf16 TT_BENCH_F32=0 TT_BENCH_TICK=0 TT_BENCH_VOXEL=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_bf16_a.json 2> gpurun_out/r2_bench_bf16_a.err; echo "bench bf16 rc=$?"
TT_DEC_FUSED_CONCAT=1 TT_BENCH_F32=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_f16_fusedcat.json 2> gpurun_out/r2_bench_f16_fusedcat.err; echo "bench fusedcat rc=$?"
for v in 6 7; do
  for shape in "64 112 224 256 256 3" "8 112 112 512 512 3" "64 56 112 256 256 3"; do
    TT_GLDS_VARIANT=$v timeout 120 python tools/conv_microbench.py $shape 1 bf16 20 2>&1 | head -1
  done
done > gpurun_out/r2_microbench_v7.txt 2>&1
cat gpurun_out/r2_microbench_v7.txt
head -c 1500 gpurun_out/r2_bench_f16_a.json
