#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_backward.py -m gpu -q -x -s -k "gru" 2>&1 | tail -25
