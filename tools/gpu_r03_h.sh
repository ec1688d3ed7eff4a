#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_train_step.py -x -q -s -k "f11 or f16" > gpurun_out/r3h_pytest_trainmode.txt 2>&1; tail -40 gpurun_out/r3h_pytest_trainmode.txt | cut -c1-250
