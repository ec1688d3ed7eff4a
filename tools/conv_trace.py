#!/usr/bin/env python
"""Where a short-K tile's time goes: per-workgroup wall-clock stamps of the LDS-DMA conv kernel (tt_conv_set_trace) on the 1x1 layers
of tools/shortk_ab.py -- prologue (entry -> first K tile landed), K loop, epilogue; medians over the workgroups, the spread of the
epilogue START times inside a round (are the CUs in lock step?), and the launch's wall time against the sum of the phases.
    python tools/conv_trace.py            (needs a GPU)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

SHAPES = [
    (64, 28, 56, 256, 1024, True),       # M = 100,352  N = 1024 K = 256
    (32, 112, 224, 256, 1280, False),    # M = 802,816  N = 1280 K = 256   decoder value GEMM
    (64, 28, 56, 1024, 256, False),      # M = 100,352  N = 256  K = 1024
    (64, 56, 112, 128, 512, True),       # M = 401,408  N = 512  K = 128
    (64, 112, 224, 64, 256, True),       # M = 1,605,632 N = 256 K = 64
    (64, 14, 28, 512, 2048, True),       # M = 25,088   N = 2048 K = 512
]


def main():
    from thinktwice_amd import _lib, ops, weights
    L = _lib.lib()
    for (N, H, W, Cin, Cout, rr) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(5)
        x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
        w = torch.randn(Cout, 1, 1, Cin, device="cuda", generator=g) * Cin ** -0.5
        r = torch.randn(N, H, W, Cout, device="cuda", generator=g) if rr else None
        wx = weights.split_pairs_x3(w)
        out = torch.empty(N, H, W, Cout, device="cuda")
        conv = lambda: ops.conv2d(x, w, act=1 if rr else 0, res1=r, w_x3=wx, out=out)
        for _ in range(3):
            conv()
        kern = ops._last_conv_kernel()
        M = N * H * W
        nblk = 65536
        stamps = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        L.tt_conv_set_trace(ctypes.c_void_p(stamps.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        conv()
        e1.record()
        torch.cuda.synchronize()
        L.tt_conv_set_trace(None)
        s = stamps.cpu().numpy().reshape(-1, 4)
        s = s[s[:, 0] > 0]
        t0 = s[:, 0].min()
        us = (s - t0) * 0.01
        pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
        med = lambda a: float(np.median(a))
        print(f"M={M} N={Cout} K={Cin}{' +res' if rr else ''}  {kern.split('>')[0]}>  {len(s)} workgroups, launch {e0.elapsed_time(e1) * 1e3:.0f} us (traced)")
        print(f"   prologue {med(pro):6.2f} us   K loop {med(loop):6.2f}   epilogue {med(epi):6.2f}   (medians; p10-p90: "
              f"{np.percentile(pro, 10):.1f}-{np.percentile(pro, 90):.1f} / {np.percentile(loop, 10):.1f}-{np.percentile(loop, 90):.1f} / "
              f"{np.percentile(epi, 10):.1f}-{np.percentile(epi, 90):.1f})   last workgroup done at {us[:, 3].max():.0f} us")
        # lock step? start times of the epilogues, in 2 us bins over the launch
        edges = np.arange(0, us[:, 3].max() + 4, 4.0)
        h_epi = np.zeros(len(edges) - 1)
        for a, b in zip(us[:, 2], us[:, 3]):
            lo, hi = int(a // 4), min(int(b // 4), len(h_epi) - 1)
            h_epi[lo:hi + 1] += 1
        print("   workgroups inside their epilogue per 4 us bin: " + " ".join(f"{int(v)}" for v in h_epi[:60]))
        del x, w, r, out, stamps
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
