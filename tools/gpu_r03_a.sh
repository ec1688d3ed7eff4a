#!/bin/bash
# round 3, call A: run-staged sparse conv -- parity tests + A/B bench with per-shape conv table
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv.py -x -q -k "sparse or gathered" > gpurun_out/r3f_pytest_conv.txt 2>&1
tail -5 gpurun_out/r3f_pytest_conv.txt
timeout 600 python -m pytest tests/test_lidar.py -x -q > gpurun_out/r3f_pytest_lidar.txt 2>&1
tail -5 gpurun_out/r3f_pytest_lidar.txt
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0
TT_SP_RUNS=1 TT_BENCH_DUMP=gpurun_out/r3f_shapes_runs1.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3f_bench_runs1.json 2> gpurun_out/r3f_bench_runs1.err
TT_SP_RUNS=0 TT_BENCH_DUMP=gpurun_out/r3f_shapes_runs0.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3f_bench_runs0.json 2> gpurun_out/r3f_bench_runs0.err
python - <<'PY'
import json
for k in (1, 0):
    try:
        b = json.load(open(f"gpurun_out/r3f_bench_runs{k}.json"))
        print("runs", k, b["value"], b["ms_per_step"], b["roofline"]["conv_ms_per_step"])
        for r in json.load(open(f"gpurun_out/r3f_shapes_runs{k}.json")):
            if "sparse" in r["shape"]:
                print("   ", r)
    except Exception as e:
        print("runs", k, "failed", e)
PY
