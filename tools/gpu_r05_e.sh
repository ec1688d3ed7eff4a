#!/bin/bash
# round 5, call E: persistent short-K kernel, 8- vs 4-wave geometry
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r05_e
timeout 900 python tools/shortk_ab.py base pers pers4 2>&1 | tee $O.shortk.txt | cut -c1-230
