#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r3_pytest_gpu_b.txt 2>&1; tail -15 gpurun_out/r3_pytest_gpu_a.txt | cut -c1-300
