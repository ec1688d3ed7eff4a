#!/bin/bash
# round 3, call G: train-mode BatchNorm / dropout kernels, F11 parity, frozen-BN regression, action-post device entry
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_batchnorm.py tests/test_control.py tests/test_agent_tick.py -x -q > gpurun_out/r3g_pytest_bn.txt 2>&1; tail -15 gpurun_out/r3g_pytest_bn.txt
timeout 900 python -m pytest tests/test_conv_bwd.py -x -q -k "zero_and_tiny or epilogue or residual_layer" > gpurun_out/r3g_pytest_convbwd.txt 2>&1; tail -5 gpurun_out/r3g_pytest_convbwd.txt
timeout 1500 python -m pytest tests/test_train_step.py -x -q -s -k "f11 or f13 or updates" > gpurun_out/r3g_pytest_train.txt 2>&1; tail -30 gpurun_out/r3g_pytest_train.txt
