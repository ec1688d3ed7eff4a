#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out
cd "$ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_train_step.py -m gpu -q -s 2>&1 | grep -v "socket.cpp\|Gloo\|amdgpu.ids" | tail -12
timeout 600 python bench.py --workload train_step --steps 3 --warmup 1 --batch 8 > $OUT/train_step_b8_dev.json 2> $OUT/train_step_b8_dev.err
grep -i "trainer\]" $OUT/train_step_b8_dev.err | head -3
python - <<PY
import json
d=json.load(open("$OUT/train_step_b8_dev.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["wgrad"], d["roofline"]["kernel_ms"], d["train_step_phases"])
PY
