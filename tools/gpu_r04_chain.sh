#!/bin/bash
# round 4: the wide form of the decoder's row chains (tt_mlp_chain_wide): tests, then the batch-1 tick and the batch-8 step with
# the form on / off (each arm its own process)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
O=gpurun_out/r04_chain_wide.txt; : > $O
timeout 900 python -m pytest tests/test_chain.py tests/test_decoder_fused.py tests/test_decoder.py -q -m gpu -x 2>&1 | tail -6 | tee -a $O
for w in 0 1 0 1; do
  echo "== TT_CHAIN_WIDE=$w" | tee -a $O
  TT_CHAIN_WIDE=$w timeout 300 python tools/tick_profile.py f32x3 10 2>&1 | grep "^tick" | tee -a $O
done
for w in 0 1; do
TT_CHAIN_WIDE=$w timeout 600 python - <<'PY' 2>&1 | tail -2 | tee -a $O
import os, json, torch
from thinktwice_amd.bench_forward import ForwardWorkload
w = ForwardWorkload(8, torch.device("cuda", 0))
for _ in range(3): w.step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): w.step()
torch.cuda.synchronize()
print("TT_CHAIN_WIDE=" + os.environ["TT_CHAIN_WIDE"], "B=8 step ms", (time.perf_counter() - t0) * 100)
PY
done
