#!/bin/bash
# round 5, call D: the training tests that failed at the checkpoint (unaligned value_proj view under the trainer), persistent short-K kernel A/B
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r05_d
rm -rf $O.*
timeout 900 python tools/shortk_ab.py base pers 2>&1 | tee $O.shortk.txt | cut -c1-220
timeout 1200 python -m pytest "tests/test_backward.py::test_decoder_backward_matches_oracle_autograd" \
   "tests/test_train_step.py::test_training_backward_matches_reference_gradients_golden_f13" \
   "tests/test_train_step.py::test_training_backward_in_train_mode_matches_reference_gradients_golden_f16" \
   -q -m gpu --maxfail=8 2>&1 | tail -40 > $O.pytest.txt
cut -c1-400 $O.pytest.txt | tail -30
