#!/usr/bin/env python
"""Time the camera-trunk layers of the mixed mode (ResNet layer1 / layer2, PAFPN at B = 8: 64 images of 448 x 896) through
tt_conv2d_fwd in their bf16x3 form (f32 storage) and in the h2 form (half storage, f16 (hi, lo) weights):

    [TT_H2_TILE=n] python tools/h2_microbench.py [x3] [h2] [--images 64] [--iters 10] [--only substr]

One line per (layer, mode): ms, algorithmic TFLOP/s, compulsory GB and TB/s, the kernel that ran."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import ops, weights  # noqa: E402

# name, H, W, Cin, Cout, k, stride, residual, f32 output (h2), calls per forward
LAYERS = [
    ("l1 1x1 64>64", 112, 224, 64, 64, 1, 1, 0, 0, 1),
    ("l1 3x3 64>64", 112, 224, 64, 64, 3, 1, 0, 0, 3),
    ("l1 1x1 64>256+res", 112, 224, 64, 256, 1, 1, 1, 0, 3),
    ("l1 ds 64>256", 112, 224, 64, 256, 1, 1, 0, 0, 1),
    ("l1 1x1 256>64", 112, 224, 256, 64, 1, 1, 0, 0, 2),
    ("l2 1x1 256>128", 112, 224, 256, 128, 1, 1, 0, 0, 1),
    ("l2 3x3s2 128>128", 112, 224, 128, 128, 3, 2, 0, 0, 1),
    ("l2 1x1 128>512+res", 56, 112, 128, 512, 1, 1, 1, 0, 4),
    ("l2 ds s2 256>512", 112, 224, 256, 512, 1, 2, 0, 0, 1),
    ("l2 1x1 512>128", 56, 112, 512, 128, 1, 1, 0, 0, 3),
    ("l2 3x3 128>128", 56, 112, 128, 128, 3, 1, 0, 0, 3),
    ("l3.0 1x1 512>256 f32out", 56, 112, 512, 256, 1, 1, 0, 1, 1),
    ("l3.0 ds s2 512>1024 f32out", 56, 112, 512, 1024, 1, 2, 0, 1, 1),
    ("lat0 1x1 256>256", 112, 224, 256, 256, 1, 1, 0, 0, 1),
    ("lat1 1x1 512>256", 56, 112, 512, 256, 1, 1, 0, 0, 1),
    ("fpn0 3x3 256>256 f32out", 112, 224, 256, 256, 3, 1, 0, 1, 1),
    ("fpn1 3x3 256>256", 56, 112, 256, 256, 3, 1, 0, 0, 1),
    ("paf0 3x3 256>256 f32out", 56, 112, 256, 256, 3, 1, 0, 1, 1),
    ("down0 3x3s2 256>256+res", 112, 224, 256, 256, 3, 2, 1, 0, 1),
    ("fpn2 3x3 256>256", 28, 56, 256, 256, 3, 1, 0, 0, 1),
    ("paf1 3x3 256>256 f32out", 28, 56, 256, 256, 3, 1, 0, 1, 1),
    ("down1 3x3s2 256>256+res", 56, 112, 256, 256, 3, 2, 1, 0, 1),
]


def main():
    args = sys.argv[1:]
    modes = [a for a in args if a in ("x3", "h2")] or ["x3", "h2"]
    NI = int(args[args.index("--images") + 1]) if "--images" in args else 64
    iters = int(args[args.index("--iters") + 1]) if "--iters" in args else 10
    only = args[args.index("--only") + 1] if "--only" in args else None
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = {m: 0.0 for m in modes}
    print(f"# images={NI} iters={iters} TT_H2_TILE={os.environ.get('TT_H2_TILE')}")
    for name, H, W, Cin, Cout, k, stride, res, f32out, calls in LAYERS:
        if only and only not in name:
            continue
        pad = k // 2
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        M = NI * OH * OW
        w32 = torch.randn(Cout, k, k, Cin, device="cuda", generator=g) * (Cin * k * k) ** -0.5
        for mode in modes:
            if mode == "x3":
                x = torch.randn(NI, H, W, Cin, device="cuda", generator=g)
                r = torch.randn(NI, OH, OW, Cout, device="cuda", generator=g) if res else None
                wx = weights.split_pairs_x3(w32)
                run = lambda: ops.conv2d(x, w32, stride=stride, pad=pad, act=1, res1=r, w_x3=wx)
                esz, osz = 4, 4
            else:
                x = torch.randn(NI, H, W, Cin, device="cuda", generator=g).half()
                r = torch.randn(NI, OH, OW, Cout, device="cuda", generator=g).half() if res else None
                wh, w16 = weights.split_pairs_h2(w32), w32.half()
                od = torch.float32 if f32out else None
                run = lambda: ops.conv2d(x, w16, stride=stride, pad=pad, act=1, res1=r, w_h2=wh, out_dtype=od)
                esz, osz = 2, (4 if f32out else 2)
            for _ in range(2):
                y = run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                y = run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            fl = 2.0 * M * Cout * k * k * Cin
            in_px = min(NI * H * W, M * k * k)
            gb = (in_px * Cin * esz + M * Cout * osz + (M * Cout * esz if res else 0)) / 1e9
            tot[mode] += ms * calls
            print(f"{name:30s} M={M:8d} N={Cout:4d} K={k * k * Cin:5d} {mode}: {ms:7.3f} ms x{calls} {fl / ms / 1e9:7.1f} TF/s "
                  f"{gb:6.3f} GB {gb / ms:6.2f} TB/s  {ops._last_conv_kernel()}", flush=True)
            del x, r, y
    for m in modes:
        print(f"# total {m}: {tot[m]:.3f} ms per forward (calls weighted)")


if __name__ == "__main__":
    main()
