#!/bin/bash
# round 6, call AA: do the wide chains' agent-scope fences (L2 write-back + invalidate per barrier) cost the co-running trunk kernels
# anything at batch 8?  Step with the wide form (default) against the per-row-block chain kernel (TT_CHAIN_WIDE=0)
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_aa.txt; rm -f $O
for w in 1 0 1 0; do
  echo "TT_CHAIN_WIDE=$w" | tee -a $O
  TT_CHAIN_WIDE=$w timeout 600 python tools/pipeline_ab.py 10 8 2>&1 | grep "in flight" | tee -a $O
done
