#!/bin/bash
# round 6, call X: timing ablations of conv_h2_pipe_kernel on the dominant PAFPN layer (results wrong by design)
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_x.txt; rm -f $O
echo "full kernel" | tee -a $O
timeout 300 python tools/conv_microbench.py 64 112 224 256 256 3 1 h2 10 2>&1 | grep "h2:" | tee -a $O
for n in 1 2 3 4 7; do
  echo "TT_H2_DEBUG=$n (1 no DMA, 2 no fragment reads, 4 no barrier)" | tee -a $O
  TT_LIB_PATH=$ROOT/tools/_dbg/libtt_h2_$n.so timeout 300 python tools/conv_microbench.py 64 112 224 256 256 3 1 h2 10 2>&1 | grep "h2:" | tee -a $O
done
