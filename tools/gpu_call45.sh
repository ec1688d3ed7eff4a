#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_backward.py -m gpu -q -s -k "decoder_backward" 2>&1 | tail -25
timeout 900 python -m pytest tests/test_train_step.py -m gpu -q -s 2>&1 | tail -60
