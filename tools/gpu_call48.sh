#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out
cd "$ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_bwd.py tests/test_train_step.py -m gpu -q -s 2>&1 | tail -8
timeout 300 python -m pytest tests/test_backward.py -m gpu -q -k "lidar or trunk" 2>&1 | tail -3
for xcd in 0 1; do
  echo "TT_WGRAD_XCD=$xcd"
  for shp in "64 112 224 64 64 3" "64 56 112 128 128 3" "64 28 56 256 256 3" "64 14 28 512 512 3" "32 224 448 32 32 3"; do
    TT_WGRAD_XCD=$xcd timeout 120 python tools/wgrad_microbench.py $shp 2>&1 | grep wgrad
  done
done
for xcd in 0 1; do
TT_WGRAD_XCD=$xcd timeout 600 python bench.py --workload train_step --steps 3 --warmup 1 --batch 8 > $OUT/train_step_b8_xcd$xcd.json 2> $OUT/train_step_b8.err
python - <<PY
import json
d=json.load(open("$OUT/train_step_b8_xcd$xcd.json"))
print("xcd=$xcd", d["value"], d["ms_per_step"], d["roofline"]["wgrad"], d["roofline"]["kernel_ms"], d["train_step_phases"])
PY
done
