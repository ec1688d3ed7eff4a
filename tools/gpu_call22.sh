#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python tools/dec_microbench.py 2>&1 | grep -v "^W\|amdgpu" | head -30
