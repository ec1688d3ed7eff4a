#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_backward.py tests/test_train_step.py -m gpu -x -q -k "lidar or f13 or f16 or updates" 2>&1 | tail -4
timeout 900 python bench.py --workload train_step --steps 3 --warmup 1 2> /dev/null | cut -c1-260
