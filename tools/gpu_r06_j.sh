#!/bin/bash
# round 6, call J: find the crash in the GPU suite and the smoke() mismatch
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_j.txt; rm -f $O
timeout 300 python tools/smoke_debug.py f32 1 2>&1 | grep -v amdgpu.ids | tail -15 | tee -a $O
timeout 300 python tools/smoke_debug.py f32 2 2>&1 | grep -v amdgpu.ids | tail -15 | tee -a $O
timeout 300 python tools/smoke_debug.py f32x3 1 2>&1 | grep -v amdgpu.ids | tail -15 | tee -a $O
timeout 2400 python -m pytest tests/ -x -v -m gpu 2>&1 | grep -v amdgpu.ids | grep -n "PASSED\|FAILED\|ERROR\|Fatal\|rror" | tail -12 | cut -c1-250 | tee -a $O
