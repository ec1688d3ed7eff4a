#!/bin/bash
# round 6, call A: h2 conv kernel -- parity tests, then the layer table (bf16x3 vs h2) with the tile variants
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_a.txt; rm -f $O
timeout 600 python -m pytest tests/test_conv.py -x -q -m gpu -k "h2 or store_half" 2>&1 | tail -15 | tee -a $O
timeout 600 python tools/h2_microbench.py x3 h2 2>&1 | tee -a $O
for t in 1 3; do TT_H2_TILE=$t timeout 300 python tools/h2_microbench.py h2 2>&1 | tee -a $O; done
TT_H2_TILE=2 timeout 300 python tools/h2_microbench.py h2 --only "l1 " 2>&1 | tee -a $O
