"""Time tt_conv2d_wgrad and the dgrad convolution on one shape:  tools/wgrad_microbench.py N H W Cin Cout k [stride]"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import ops  # noqa: E402

N, H, W, Cin, Cout, k = (int(v) for v in sys.argv[1:7])
stride = int(sys.argv[7]) if len(sys.argv) > 7 else 1
pad = k // 2
OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
x = torch.randn(N, H, W, Cin, device="cuda")
dy = torch.randn(N, OH, OW, Cout, device="cuda")
w = torch.randn(Cout, k, k, Cin, device="cuda") * (Cin * k * k) ** -0.5


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


fl = 2.0 * N * OH * OW * Cout * k * k * Cin
t = timeit(lambda: ops.conv2d_wgrad(x, dy, k, k, stride, pad, 1))
print(f"wgrad M={N * OH * OW} N={Cout} K={k * k * Cin} s{stride}: {t:.3f} ms  {fl / t / 1e9:.1f} TF/s (f32 MFMA peak 157.3)")
t = timeit(lambda: ops.conv2d_dgrad(dy, w, (H, W), stride, pad, 1, x3=True))
print(f"dgrad (bf16x3 forward kernel on rotated weights, incl. the weight transform): {t:.3f} ms  {fl / t / 1e9:.1f} TF/s")
