#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_backward.py tests/test_conv_bwd.py tests/test_losses.py -m gpu -q 2>&1 | tail -5
