"""Forward conv / dgrad / wgrad / epilogue-backward on tiny maps with several images (debug tool)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import ops
torch.manual_seed(0)
for (N, H, W, Cin, Cout, k, s) in [(8, 2, 4, 512, 512, 3, 1), (8, 2, 4, 2048, 512, 1, 1), (8, 2, 4, 512, 2048, 1, 1), (4, 16, 32, 64, 256, 1, 1),
                                   (4, 16, 32, 256, 64, 1, 1), (4, 16, 32, 64, 64, 3, 1), (8, 4, 8, 1024, 256, 1, 1), (8, 4, 8, 256, 256, 3, 1)]:
    pad = k // 2
    x = torch.randn(N, Cin, H, W, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k) * (Cin * k * k) ** -0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, s, pad)
    dy = torch.randn_like(y)
    y.backward(dy)
    cl = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()
    xq, wq, dyq = cl(x), cl(w), cl(dy)
    yf = ops.conv2d(xq, wq, stride=s, pad=pad)
    dx = ops.conv2d_dgrad(dyq, wq, (H, W), s, pad, 1, x3=False)
    dw = ops.conv2d_wgrad(xq, dyq, k, k, s, pad, 1)
    _, _, dsc, dsh = ops.conv_epilogue_bwd(dyq, yf, torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda"), 1)
    mask = (y.detach() > 0).float()
    want_dsh = (dy * mask).sum((0, 2, 3))
    rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
    print(f"N={N} {H}x{W} {Cin}->{Cout} k{k}: fwd {rel(yf.permute(0,3,1,2), y.detach()):.1e} dgrad {rel(dx.permute(0,3,1,2), x.grad):.1e} "
          f"wgrad {rel(dw.permute(0,3,1,2), w.grad):.1e} dshift {rel(dsh, want_dsh):.1e}")
