#!/usr/bin/env python
"""A/B of the PAFPN top-down path at B = 8 (64 images): lateral conv + tt_upsample_nearest_add against the lateral conv with
the upsampled residual in its epilogue (tt_conv_desc.res1_up_*), levels 0-2; bit-equal outputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import ops, weights  # noqa: E402


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = [0.0, 0.0]
    for (H, W, Cin) in ((112, 224, 256), (56, 112, 512), (28, 56, 1024)):
        x = torch.randn(64, H, W, Cin, device="cuda", generator=g)
        w = torch.randn(256, 1, 1, Cin, device="cuda", generator=g) * Cin ** -0.5
        b = torch.randn(256, device="cuda", generator=g)
        wx = weights.split_pairs_x3(w)
        coarse = torch.randn(64, H // 2, W // 2, 256, device="cuda", generator=g)
        out = torch.empty(64, H, W, 256, device="cuda")

        def sep():
            ops.conv2d(x, w, shift=b, w_x3=wx, out=out)
            ops.upsample_nearest_add_(out, coarse)

        def fused():
            ops.conv2d(x, w, shift=b, w_x3=wx, out=out, res1=coarse, res1_up=True)
        sep()
        a = out.clone()
        fused()
        same = torch.equal(a, out)
        t0, t1 = timed(sep), timed(fused)
        tot[0] += t0
        tot[1] += t1
        print(f"level {H}x{W} Cin={Cin}: conv + upsample_add {t0:.3f} ms, fused {t1:.3f} ms, bit-equal {same}  [{ops._last_conv_kernel()}]", flush=True)
    print(f"total: {tot[0]:.3f} -> {tot[1]:.3f} ms per forward")


if __name__ == "__main__":
    main()
