"""The batch-1 tick issued from the C launch plan under a profiler (tools only; the counterpart of tools/tick_profile.py):
   rocprofv3 --kernel-trace -- python tools/tick_profile_plan.py [iters]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from thinktwice_amd import model as tm, params, plan as P, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m, cfg = tm.build_thinktwice(dtype="f32x3")
m.load_state_dict(params.init_params(cfg, seed=0))
b1 = tm.batch_to_device(synth.make_batch(1, seed=4321))
fp = P.compile_forward(m, b1, channel_last_out=True)
for _ in range(2):
    fp.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    fp.run()
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
time.sleep(0.05)                     # idle gap: tools/last_tick_stats.py cuts the trace here
fp.run()
torch.cuda.synchronize()
print(f"tick from the C plan: {dt * 1e3:.2f} ms ({fp.calls} calls, {fp.nstreams} streams, arena {fp.arena.numel() / 1e9:.2f} GB)")
