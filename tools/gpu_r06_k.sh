#!/bin/bash
# round 6, call K: whole GPU suite (verbose tail) + smoke() in the driver's order, then the step time
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_k.txt; rm -f $O
timeout 3000 python -m pytest tests/ -v -m gpu 2>&1 | grep -v amdgpu.ids > $ROOT/gpurun_out/r06_k_pytest_full.txt
grep -c PASSED $ROOT/gpurun_out/r06_k_pytest_full.txt | sed 's/^/PASSED: /' | tee -a $O
grep "FAILED\|ERROR\|Fatal\|core\|passed\|failed" $ROOT/gpurun_out/r06_k_pytest_full.txt | tail -12 | cut -c1-300 | tee -a $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O
timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | head -3 | tee -a $O
