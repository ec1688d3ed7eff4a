"""Closed-loop tick (batch 1; or one forward of [batch] frames) under a profiler (tools only):
rocprofv3 --kernel-trace --stats -- python tools/tick_profile.py [dtype] [iters] [batch]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from thinktwice_amd import model as tm, params, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}.get(mode, mode)
m, cfg = tm.build_thinktwice(dtype=dt)
m.load_state_dict(params.init_params(cfg, seed=0))
b1 = tm.batch_to_device(synth.make_batch(batch, seed=4321))
for _ in range(2):
    m.forward_inference(b1, channel_last_out=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    m.forward_inference(b1, channel_last_out=True)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
time.sleep(0.05)                     # idle gap: tools/last_tick_stats.py cuts the trace here
m.forward_inference(b1, channel_last_out=True)
torch.cuda.synchronize()
t0 = time.perf_counter() - dt * iters
print(f"tick {mode}: {(time.perf_counter() - t0) / iters * 1e3:.2f} ms")
