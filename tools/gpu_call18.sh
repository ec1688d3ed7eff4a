#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_forward.py -m gpu -q -x -k "train_losses" -s 2>&1 | tail -25
