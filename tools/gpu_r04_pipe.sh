#!/bin/bash
# round 4: A/B of the hand-pipelined 4-wave bf16x3 tile against the 8-wave tile (bit-exactness + timing)
mkdir -p gpurun_out
timeout 900 python tools/x3_pipe_ab.py 2 ${ARMS:-0,1,2} > gpurun_out/r04_pipe_ab.txt 2>&1
echo "rc=$?" >> gpurun_out/r04_pipe_ab.txt
tail -20 gpurun_out/r04_pipe_ab.txt
