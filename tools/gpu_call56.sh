#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out
cd "$ROOT" || exit 1
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_forward.py -m gpu -q -k "f10" 2>&1 | tail -3
timeout 600 python bench.py --workload train_step --steps 2 --warmup 1 --batch 8 > $OUT/train_step_b8_h.json 2> $OUT/train_step_b8_h.err
tail -3 $OUT/train_step_b8_h.err
python - <<PY
import json
d=json.load(open("$OUT/train_step_b8_h.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["wgrad"], d["train_step_phases"])
PY
