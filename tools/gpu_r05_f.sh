#!/bin/bash
# round 5, call F: the training-step leg after the aligned value_proj copy (device-side operand preparation back), decoder backward tests, the
# persistent-kernel test
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r05_f
timeout 600 python bench.py --workload train_step --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-1200 | tee $O.train.txt
timeout 900 python -m pytest "tests/test_backward.py::test_decoder_backward_matches_oracle_autograd" "tests/test_conv.py::test_persistent_shortk_gemm_is_bit_identical_to_the_8wave_tile" tests/test_train_step.py -q -m gpu --maxfail=5 2>&1 | tail -8 | cut -c1-300 | tee $O.pytest.txt
