#!/bin/bash
# round 6, call B: bench step with 1 / 2 / 3 batches in flight
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_b.txt; rm -f $O
timeout 900 python tools/pipeline_ab.py 10 8 2>&1 | grep -v amdgpu.ids | tee -a $O
