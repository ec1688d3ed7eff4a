#!/bin/bash
# round 4: training-step tests (gradient goldens, BatchNorm, trunk backward) + the batch-8 iteration with the tape releasing
# activations / gradient buffers during the sweep (TT_TAPE_EAGER_RELEASE=1, default) and retaining them (=0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
O=gpurun_out/r04_train.txt; : > $O
timeout 1500 python -m pytest tests/test_train_step.py tests/test_batchnorm.py "tests/test_backward.py::test_camera_trunk_backward_matches_oracle_autograd" -q -m gpu -x 2>&1 | tail -5 | tee -a $O
for r in ${ARMS:-1 0}; do
  echo "== TT_TAPE_EAGER_RELEASE=$r" | tee -a $O
  TT_TAPE_EAGER_RELEASE=$r timeout 600 python bench.py --workload train_step --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(json.dumps({k:d[k] for k in ('value','unit','ms_per_step') if k in d}), json.dumps(d.get('train_step_phases')))" | tee -a $O
done
