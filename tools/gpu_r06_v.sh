#!/bin/bash
# round 6, call V: rocprofv3 kernel stats of the training iteration (B = 8)
ROOT="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r6train -o p -- python $ROOT/bench.py --workload train_step --steps 2 --warmup 1 > $ROOT/gpurun_out/r06_v_train.json 2> $ROOT/gpurun_out/r06_v_train.err
cd $ROOT
f=$(find gpurun_out/r6train -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r06_train_step_kernel_stats.csv
rm -rf gpurun_out/r6train
tail -c 600 gpurun_out/r06_v_train.json
