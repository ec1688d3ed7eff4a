#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp TT_LIB_PATH=$PWD/tools/_dbg/libthinktwice_hip.so TT_GLDS_X3_PINGPONG=1
for act in 0 99 96 95 97 98 94 93; do
  TT_MB_ACT=$act timeout 120 python tools/conv_microbench.py 64 112 224 256 256 3 1 x3 10 2>&1 | grep "M=" | cut -c1-60 | sed "s/^/pp act=$act /"
done
