#!/bin/bash
# round 5, call E: persistent short-K kernel A/B (tools/shortk_ab.py arms given in ARMS)
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r05_e
timeout 900 python tools/shortk_ab.py ${ARMS:-base pers} 2>&1 | tee $O.shortk.txt | cut -c1-230
