#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python tools/small_map_conv_check.py 2>&1 | tail -18
