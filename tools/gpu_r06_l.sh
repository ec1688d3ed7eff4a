#!/bin/bash
# round 6, call L: top-down fusion A/B; pipeline depth 3 / 4 / 5
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_l.txt; rm -f $O
timeout 300 python tools/upadd_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O
timeout 300 python -m pytest tests/test_conv.py tests/test_forward.py -x -q -m gpu -k "upsampled or in_flight" 2>&1 | tail -3 | tee -a $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $O
import sys, time, torch
sys.path.insert(0, ".")
from thinktwice_amd import bench_forward
wl = bench_forward.ForwardWorkload(8, torch.device("cuda", 0))
for depth in (3, 4, 5, 3, 1):
    wl.pipeline, wl._streams, wl._tick = depth, None, 0
    for _ in range(depth + 1):
        wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        wl.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 12 * 1e3
    print(f"batches in flight {depth}: {ms:.2f} ms per step = {8 / ms * 1e3:.2f} frames/s  (allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak)", flush=True)
PY
