#!/bin/bash
# round 3 final: the whole -m gpu suite, then smoke()
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | tail -25 > gpurun_out/r3_pytest_gpu_final.txt
cat gpurun_out/r3_pytest_gpu_final.txt | cut -c1-400
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
