#!/usr/bin/env python
"""Time one conv shape through tt_conv2d_fwd:  tools/conv_microbench.py N H W Cin Cout k [stride] [dtype] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import ops  # noqa: E402


def main():
    a = sys.argv[1:]
    N, H, W, Cin, Cout, k = (int(v) for v in a[:6])
    stride = int(a[6]) if len(a) > 6 else 1
    mode = a[7] if len(a) > 7 else "bf16"
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}.get(mode, torch.float32)
    iters = int(a[8]) if len(a) > 8 else 20
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g).to(dt)
    w = (torch.randn(Cout, k, k, Cin, device="cuda", generator=g) * (Cin * k * k) ** -0.5).to(dt)
    pad = k // 2
    act = int(os.environ.get("TT_MB_ACT", "0"))
    res = torch.randn(N, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1, Cout, device="cuda").to(dt) \
        if os.environ.get("TT_MB_RES") else None
    from thinktwice_amd import weights
    wx = weights.split_pairs_x3(w) if mode in ("x3", "x3p") else None
    if mode == "x3p":       # pre-split (pair-format) activations: tt_conv_desc.in_pair
        x = weights.split_pairs_x3(x)
    if mode == "h2":        # half activations x f16 (hi, lo) weights (csrc/conv_h2.hip), f32 output
        wh, x, w = weights.split_pairs_h2(w), x.half(), w.half()
        conv = lambda: ops.conv2d(x, w, stride=stride, pad=pad, act=act, w_h2=wh, out_dtype=torch.float32)
    else:
        conv = lambda: ops.conv2d(x, w, stride=stride, pad=pad, act=act, res1=res, w_x3=wx, in_pair=mode == "x3p")
    for _ in range(3):
        y = conv()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = conv()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    M = y.shape[0] * y.shape[1] * y.shape[2]
    fl = 2.0 * M * Cout * k * k * Cin
    print(f"M={M} N={Cout} K={k*k*Cin} {mode}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF/s  ACT={act} RES={res is not None}  {ops._last_conv_kernel()}")
    # reference point: a plain device copy of the output-sized tensor (read + write M*N elements)
    src = torch.empty_like(y)
    for _ in range(3):
        src.copy_(y)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        src.copy_(y)
    e1.record()
    torch.cuda.synchronize()
    cms = e0.elapsed_time(e1) / iters
    print(f"   copy of the output tensor ({y.numel() * y.element_size() / 1e6:.0f} MB): {cms:.3f} ms = {2 * y.numel() * y.element_size() / cms / 1e9:.2f} TB/s")


if __name__ == "__main__":
    main()
