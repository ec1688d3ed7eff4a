#!/bin/bash
# round 4: run-staged (RUN3) 3x3 tiles against the per-tap pipelined tiles and the 8-wave tiles; bit-exactness + timing
mkdir -p gpurun_out
O=gpurun_out/r04_run3_ab.txt; : > $O
echo "== 256-wide: arm0 = 8-wave tile, arm2 = pipelined tile; TT_X3_RUN3=1 (run-staged where the layer allows)" >> $O
TT_X3_RUN3=1 timeout 600 python tools/x3_pipe_ab.py 2 0,2 >> $O 2>&1
echo "== 256-wide, TT_X3_RUN3=0 (per-tap pipelined tile everywhere)" >> $O
TT_X3_RUN3=0 timeout 600 python tools/x3_pipe_ab.py 2 0,2 >> $O 2>&1
echo "== 128-wide: arm0 = 8-wave tile, arm1 = pipelined tile; TT_X3_RUN3=1" >> $O
TT_AB_SET=128 TT_X3_RUN3=1 timeout 600 python tools/x3_pipe_ab.py 2 0,1 >> $O 2>&1
echo "== 128-wide, TT_X3_RUN3=0" >> $O
TT_AB_SET=128 TT_X3_RUN3=0 timeout 600 python tools/x3_pipe_ab.py 2 0,1 >> $O 2>&1
grep -v "amdgpu.ids" $O | cut -c1-150
