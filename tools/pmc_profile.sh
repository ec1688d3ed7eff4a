#!/bin/bash
# Collect rocprofv3 evidence for the bench workloads on the GPU box (run from the repo root through gpurun).
# Counter passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa traces).
#   tools/pmc_profile.sh <tag>      -> gpurun_out/prof_<tag>_{trace,fetch,write}_{voxel,fwd}/
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, extra rocprof args..., -- bench args
  local name=$1; shift
  timeout 300 rocprofv3 "$@" > "$OUT/prof_${TAG}_${name}.log" 2>&1
}
V="python $ROOT/bench.py --workload voxel_pool --steps 5 --warmup 2 --no-cpu-baseline"
F="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run trace_voxel --kernel-trace --stats -d "$OUT/prof_${TAG}_trace_voxel" -o p --output-format csv -- $V
run fetch_voxel --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_${TAG}_fetch_voxel" -o p --output-format csv -- $V
run write_voxel --kernel-trace --pmc WRITE_SIZE -d "$OUT/prof_${TAG}_write_voxel" -o p --output-format csv -- $V
run trace_fwd --kernel-trace --stats -d "$OUT/prof_${TAG}_trace_fwd" -o p --output-format csv -- $F
run fetch_fwd --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_${TAG}_fetch_fwd" -o p --output-format csv -- $F
run write_fwd --kernel-trace --pmc WRITE_SIZE -d "$OUT/prof_${TAG}_write_fwd" -o p --output-format csv -- $F
ls "$OUT"/prof_${TAG}_* | head -40
