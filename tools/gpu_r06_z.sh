#!/bin/bash
# round 6, call Z: the bench line's tick leg only (eager / graph / C plan / cached)
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_X3=0 TT_BENCH_SERIAL=0 TT_BENCH_H2D=0 TT_BENCH_RAW=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_z.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_z.json"))
print(d["value"], d["ms_per_step"])
print({k: v for k, v in d["tick_latency"].items() if "ms" in k})
PY
