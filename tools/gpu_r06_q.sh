#!/bin/bash
# round 6, call Q: the 80 ms outlier of the roofline pass (M=100352 N=256 K=2304 stride 2)
ROOT="$GRAFT_REPO_ROOT"; cd $ROOT; mkdir -p gpurun_out; O=$ROOT/gpurun_out/r06_q.txt; rm -f $O
timeout 200 python tools/conv_microbench.py 64 56 112 256 256 3 2 x3 2>/dev/null | grep "^M=" | cut -c1-150 | tee -a $O
timeout 200 python tools/conv_microbench.py 64 56 112 256 256 3 2 x3p 2>/dev/null | grep "^M=" | cut -c1-150 | tee -a $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $O
import sys, torch
sys.path.insert(0, ".")
from thinktwice_amd import bench_forward
wl = bench_forward.ForwardWorkload(8, torch.device("cuda", 0))
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
for rep in range(3):
    r = wl.roofline()
    print(rep, "conv_ms", r["conv_ms_per_step"], "slowest", [(s["shape"], s["ms"]) for s in r["slowest_launches"][:3]], flush=True)
PY
