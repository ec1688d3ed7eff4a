#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_losses.py -m gpu -q -x 2>&1 | tail -15
