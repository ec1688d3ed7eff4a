#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_backward.py -m gpu -q -s -k "decoder_backward" 2>&1 | tail -5
timeout 600 python bench.py --workload train_step --steps 3 --warmup 1 --batch 8 > gpurun_out/train_step_b8.json 2> gpurun_out/train_step_b8.err
tail -5 gpurun_out/train_step_b8.err; cat gpurun_out/train_step_b8.json
