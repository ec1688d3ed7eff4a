#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out
cd "$ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_bwd.py -m gpu -q 2>&1 | tail -3
for shp in "32 224 448 32 32 3" "32 224 448 64 32 3" "64 112 224 64 256 1" "64 224 448 3 64 7 2"; do
  timeout 120 python tools/wgrad_microbench.py $shp 2>&1 | grep wgrad
done
timeout 600 python bench.py --workload train_step --steps 3 --warmup 1 --batch 8 > $OUT/train_step_b8_wide2.json 2> $OUT/train_step_b8_wide2.err
python - <<PY
import json
d=json.load(open("$OUT/train_step_b8_wide2.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["wgrad"], d["roofline"]["kernel_ms"], d["train_step_phases"])
for r in d["wgrad_top_shapes"]: print(r)
PY
