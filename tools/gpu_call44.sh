#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_backward.py -m gpu -q -s -k "look_module or decoder_backward" 2>&1 | tail -40
