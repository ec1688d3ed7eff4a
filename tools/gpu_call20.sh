#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp TT_LIB_PATH=$PWD/tools/_dbg/libthinktwice_hip.so
for act in 0 99 96 95 94 97 98; do
  TT_MB_ACT=$act timeout 120 python tools/conv_microbench.py 64 112 224 256 256 3 1 x3 10 2>&1 | grep "M=" | cut -c1-80 | sed "s/^/act=$act /"
done
for act in 0 96 94; do
  TT_MB_ACT=$act timeout 120 python tools/conv_microbench.py 64 224 448 64 64 3 1 x3 10 2>&1 | grep "M=" | cut -c1-80 | sed "s/^/act=$act /"
done
