#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out
cd "$ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_conv_bwd.py -m gpu -q -k "gather" 2>&1 | tail -8
timeout 300 python -m pytest tests/test_backward.py tests/test_train_step.py -m gpu -q -k "lidar or f13" 2>&1 | grep -v "socket.cpp\|Gloo\|amdgpu.ids" | tail -4
timeout 600 python bench.py --workload train_step --steps 3 --warmup 1 --batch 8 > $OUT/train_step_b8_g.json 2> $OUT/train_step_b8_g.err
python - <<PY
import json
d=json.load(open("$OUT/train_step_b8_g.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["wgrad"], d["roofline"]["kernel_ms"], d["train_step_phases"])
PY
