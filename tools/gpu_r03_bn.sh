#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_batchnorm.py tests/test_lss.py -m gpu -x -q 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_train_step.py tests/test_backward.py -m gpu -x -q -k "f11 or f16 or f13 or train" 2>&1 | tail -4
timeout 900 python bench.py --workload train_step --steps 3 --warmup 1 2> /dev/null | cut -c1-260
