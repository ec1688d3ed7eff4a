#!/bin/bash
# Round-2 evidence run (through gpurun): full GPU tests, default bench line, rocprofv3 kernel stats and HBM counter passes.
#   tools/gpu_profile_r02.sh <tag>    -> gpurun_out/<tag>_*
TAG=${1:-r02}
ROOT="$GRAFT_REPO_ROOT"
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd $ROOT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/${TAG}_pytest.log
TT_BENCH_DUMP=$OUT/${TAG}_conv_shapes.json timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
cd /tmp
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0
B="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
for mode in bf16x3 bf16; do
  TT_BENCH_DTYPE=$mode timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_trace_$mode" -o p --output-format csv -- $B > "$OUT/${TAG}_trace_$mode.log" 2>&1
  find "$OUT/${TAG}_trace_$mode" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/${TAG}_forward_${mode}_kernel_stats.csv"
  find "$OUT/${TAG}_trace_$mode" -name "*kernel_trace.csv" -delete
  for c in FETCH_SIZE WRITE_SIZE; do
    TT_BENCH_DTYPE=$mode timeout 400 rocprofv3 --kernel-trace --pmc $c -d "$OUT/${TAG}_pmc_${mode}_$c" -o p --output-format csv -- $B > "$OUT/${TAG}_pmc_${mode}_$c.log" 2>&1
  done
  python $ROOT/tools/summarize_pmc.py "$OUT/${TAG}_pmc_${mode}_FETCH_SIZE" "$OUT/${TAG}_pmc_${mode}_WRITE_SIZE" > "$OUT/${TAG}_forward_${mode}_pmc.json"
  find "$OUT" -path "*${TAG}_pmc_${mode}_*" -name "*.csv" -size +8M -delete
done
head -c 900 $OUT/${TAG}_bench.json; echo
ls -la $OUT | grep ${TAG}_ | head -30
