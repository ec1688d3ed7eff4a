#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for i in 1 2 3; do timeout 600 python -m pytest tests/test_backward.py -m gpu -q -s -k "whole" 2>&1 | grep "camera encoder backward\|passed\|failed"; done
timeout 600 python -m pytest tests/test_backward.py -m gpu -q -s -k "mlp_building" 2>&1 | tail -12
