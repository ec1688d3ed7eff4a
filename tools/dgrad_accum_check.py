"""Check conv2d_dgrad's accumulate-into-window path against dgrad + add over several sizes (debug tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import ops
torch.manual_seed(0)
for x3 in (False, True):
    for (N, H, W, Cin, Cout, k, s) in [(2, 16, 32, 64, 64, 3, 1), (4, 16, 32, 64, 64, 3, 1), (8, 16, 32, 64, 256, 1, 1), (8, 2, 4, 512, 512, 3, 1),
                                       (8, 2, 4, 2048, 512, 1, 1), (8, 4, 8, 256, 1024, 1, 1), (8, 8, 16, 128, 128, 3, 2), (4, 32, 64, 4, 64, 7, 2)]:
        pad = k // 2
        OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        dy = torch.randn(N, OH, OW, Cout, device="cuda")
        w = torch.randn(Cout, k, k, Cin, device="cuda") * (Cin * k * k) ** -0.5
        base = torch.randn(N, H, W, Cin + 8, device="cuda")
        want = base.clone()
        want[..., 4:4 + Cin] += ops.conv2d_dgrad(dy, w, (H, W), s, pad, 1, x3=x3)
        got = base.clone()
        ops.conv2d_dgrad(dy, w, (H, W), s, pad, 1, x3=x3, out=got, out_coff=4)
        torch.cuda.synchronize()
        err = float((got - want).abs().max() / want.abs().max())
        print(f"x3={x3} N={N} H={H} W={W} Cin={Cin} Cout={Cout} k={k} s={s}: accumulate err {err:.2e}")
