#!/bin/bash
# round 4: where the hand-pipelined bf16x3 K loop spends its time (ablation builds of tools/build_pipe_debug.sh)
mkdir -p gpurun_out
O=gpurun_out/r04_pipe_ablation.txt
: > $O
export TT_GLDS_X3_TILE=256
for shape in "64 112 224 256 256 3" "64 28 56 512 512 3"; do
  echo "== shape $shape (x3; TT_MB_ACT=99 = no epilogue)" >> $O
  for act in 0 99; do
    echo "-- 8-wave tile, act=$act" >> $O
    TT_X3_PIPE=0 TT_MB_ACT=$act python tools/conv_microbench.py $shape 1 x3 20 2>&1 | grep "TF/s" >> $O
    echo "-- pipe (product build), act=$act" >> $O
    TT_X3_PIPE=1 TT_MB_ACT=$act python tools/conv_microbench.py $shape 1 x3 20 2>&1 | grep "TF/s" >> $O
  done
  for n in 1 2 4 8 16 3 11 15; do
    echo "-- pipe debug $n (1 no DMA, 2 no split, 4 no MFMA, 8 no fragment reads, 16 no barrier), act=99" >> $O
    TT_LIB_PATH=$PWD/tools/_dbg/libtt_pipe$n.so TT_X3_PIPE=1 TT_MB_ACT=99 python tools/conv_microbench.py $shape 1 x3 20 2>&1 | grep "TF/s" >> $O
  done
done
cat $O
