"""Rare-stall check of the two-workgroup conv-GRU hand-over (tools only): per-launch times of many back-to-back tt_dec_gru
launches, alone and with a large convolution running on another stream; prints the distribution and any launch over 1 ms."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import config, decoder_fused as DF, ops, params, weights  # noqa: E402

cfg = config.model_config(final_dim=(128, 256))
sd = params.init_params(cfg, seed=0, parts=("fusion", "decoder"))
w = DF.prep_gru(sd, "decoder.decoder_layers.0.prediction_module.spatial_gru", "cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
x = torch.randn(64, 56, 112, 256, device="cuda")
wt = torch.randn(256, 3, 3, 256, device="cuda") * 0.02
wx = weights.split_pairs_x3(wt)
side = torch.cuda.Stream()
for B in (1, 8):
    inp6, state, fut = torch.randn(B, 4, 6).cuda(), torch.randn(B, 441, 32).cuda(), torch.empty(B, 4, 441, 32).cuda()
    for load in (False, True):
        for _ in range(10):
            ops.dec_gru(w, inp6, state, fut)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
        for i in range(N):
            if load and i % 4 == 0:
                with torch.cuda.stream(side):
                    ops.conv2d(x, wt, pad=1, w_x3=wx)          # ~1.2 ms of full-chip work on another stream
            ev[i][0].record()
            ops.dec_gru(w, inp6, state, fut)
            ev[i][1].record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        slow = [v for v in t if v > 1000.0]
        print(f"B={B} {'under a co-running conv' if load else 'alone':24s}: median {t[N // 2]:7.1f} us  p99 {t[int(N * 0.99)]:7.1f}  max {t[-1]:8.1f}  "
              f"launches over 1 ms: {len(slow)}  faults {ops.device_faults()}")
