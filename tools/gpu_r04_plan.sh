#!/bin/bash
# round 4: launch-plan tests + the batch-1 tick (eager / graph / C plan with and without the liveness-placed arena)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_plan.py -q -m gpu -x -s 2>&1 | grep -E "passed|failed|Error|error|plan vs|F8 through" | tail -8 | tee gpurun_out/r04_plan.txt
for c in 1 0; do
TT_PLAN_COMPACT=$c python - <<'PY' 2>&1 | tail -3 | tee -a gpurun_out/r04_plan.txt
import os, json, torch
from thinktwice_amd.bench_forward import ForwardWorkload
w = ForwardWorkload(1, torch.device("cuda", 0))
t = w.tick_latency()
print("TT_PLAN_COMPACT=" + os.environ["TT_PLAN_COMPACT"], json.dumps(t))
from thinktwice_amd import plan as P, model as tm, synth
b1 = tm.batch_to_device(synth.make_batch(1, seed=4321))
fp = P.compile_forward(w.model, b1, channel_last_out=True)
print("arena bytes", fp.arena.numel(), "recorded (bump)", fp.recorded_arena_bytes)
PY
done
