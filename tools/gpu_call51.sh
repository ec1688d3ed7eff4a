#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out
cd "$ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_bwd.py -m gpu -q 2>&1 | tail -3
timeout 300 python -m pytest tests/test_backward.py -m gpu -q -k "lidar" 2>&1 | tail -2
for shp in "64 112 224 64 64 3" "64 56 112 128 128 3" "64 28 56 256 256 3" "64 14 28 512 512 3" "64 56 112 256 128 1" "64 56 112 128 256 1" "64 28 56 1024 256 1" "64 14 28 512 2048 1"; do
  timeout 120 python tools/wgrad_microbench.py $shp 2>&1 | grep wgrad
done
timeout 600 python bench.py --workload train_step --steps 3 --warmup 1 --batch 8 > $OUT/train_step_b8_wide.json 2> $OUT/train_step_b8_wide.err
python - <<PY
import json
d=json.load(open("$OUT/train_step_b8_wide.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["wgrad"], d["roofline"]["kernel_ms"], d["train_step_phases"])
PY
