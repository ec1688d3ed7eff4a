#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for ni in 4 8; do TT_TEST_NI=$ni timeout 900 python -m pytest tests/test_backward.py -m gpu -q -x -s -k "trunk" 2>&1 | grep "trunk backward\|passed\|failed\|Error" | head -5; done
