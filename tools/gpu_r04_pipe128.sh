#!/bin/bash
# round 4: A/B of the hand-pipelined 256 x 128 tile (four 64 x 128 waves) against the 8-wave 128-wide tile
mkdir -p gpurun_out
TT_AB_SET=128 timeout 900 python tools/x3_pipe_ab.py 2 0,1 > gpurun_out/r04_pipe128_ab.txt 2>&1
echo "rc=$?" >> gpurun_out/r04_pipe128_ab.txt
tail -12 gpurun_out/r04_pipe128_ab.txt | cut -c1-260
