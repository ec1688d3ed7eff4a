#!/bin/bash
# round 4: the whole -m gpu suite (or a subset: ARGS), then smoke()
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 3300 python -m pytest ${ARGS:-tests/} -q -m gpu -x --durations=8 2>&1 | tail -30 > gpurun_out/r04_pytest_gpu.txt
cut -c1-300 gpurun_out/r04_pytest_gpu.txt
if [ -z "$ARGS" ]; then timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a gpurun_out/r04_pytest_gpu.txt; fi
