#!/bin/bash
# round 6, call P: the measurement set again on the final build (profiles + default bench line) + the train-step workload alone
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT
STAGE=prof bash tools/gpu_r06_final.sh
STAGE=bench bash tools/gpu_r06_final.sh
cd $ROOT; timeout 900 python bench.py --workload train_step --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-900 > gpurun_out/r06_train_step_alone.json; cut -c1-400 gpurun_out/r06_train_step_alone.json
