#!/bin/bash
# round 6, call U: column-split grid2feat tail -- plan / forward tests and the default bench line
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_u.txt; rm -f $O
timeout 1500 python -m pytest tests/test_plan.py tests/test_forward.py tests/test_decoder_fused.py -x -q -m gpu 2>&1 | tail -5 | tee -a $O
timeout 900 python bench.py > gpurun_out/r06_u_bench.json 2> gpurun_out/r06_u_bench.err; tail -c 1500 gpurun_out/r06_u_bench.json | tee -a $O
