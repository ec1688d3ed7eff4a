"""Convolution backward: tt_conv2d_wgrad and the input gradient through tt_conv2d_fwd on rotated weights,
against torch autograd of F.conv2d on the CPU (what the reference's loss.backward() computes for every Conv2d)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 20, 24, 64, 64, 3, 1, 1, 1),
    (3, 17, 23, 32, 96, 3, 1, 1, 1),       # odd width (ragged pixel pair), Cout tail of the 64-wide tile
    (2, 28, 28, 128, 64, 1, 1, 0, 1),      # 1x1
    (2, 30, 32, 64, 128, 3, 2, 1, 1),      # stride 2 (ResNet downsampling 3x3)
    (2, 30, 32, 64, 128, 1, 2, 0, 1),      # stride-2 1x1 (ResNet downsample branch)
    (2, 24, 24, 64, 64, 3, 1, 6, 6),       # ASPP dilation 6
    (4, 40, 48, 3, 64, 7, 2, 3, 1),        # stem: Cin = 3 (padded to 4 in the weight layout)
    (1, 21, 21, 256, 256, 3, 1, 1, 1),     # BEV-level layer: few pixels, many channels
    (8, 64, 96, 64, 64, 3, 1, 1, 1),       # many rows: split partial sums
    (2, 16, 21, 160, 200, 3, 1, 1, 1),     # wide (128-channel) wave tiles with channel tails on both sides, odd width
    (2, 14, 14, 512, 96, 1, 1, 0, 1),      # wide on the input side only
    (3, 18, 16, 48, 320, 3, 2, 1, 1),      # wide on the output side only, stride 2
    (2, 24, 31, 32, 12, 3, 1, 1, 1),       # 32-channel wave tiles on both sides (segmentation head), Cout tail, odd width
    (2, 12, 16, 24, 256, 1, 1, 0, 1),      # narrow input, wide output
    (2, 12, 16, 200, 20, 3, 1, 1, 1),      # wide input, narrow output
    # >= 128 channels on both sides: the LDS-staged bf16x3 kernel under x3 (128- / 256-channel workgroup tiles a side)
    (2, 30, 40, 128, 512, 3, 2, 1, 1),     # stride 2, 256-wide dy tile x 128-wide x tile
    (2, 20, 70, 384, 128, 1, 1, 0, 1),     # 1x1, three 32-pixel segments per row with a tail, x tile 256 wide with a channel tail
    (2, 24, 24, 256, 128, 3, 1, 6, 6),     # dilation 6: most tap rows / pixels fall outside the image
    (3, 9, 33, 256, 512, 3, 1, 1, 1),      # 256 x 256 tiles, several tiles a side, rows split over workgroups
    (3840, 1, 1, 1544, 512, 1, 1, 0, 1),   # a linear layer over rows (OW = 1): 1x1 pixels regrouped into pseudo-rows
    (40, 8, 8, 128, 128, 3, 1, 1, 1),      # rows shorter than 16 pixels and not 1x1: stays on the f32 kernels under x3 too
]


def _ref(x, w, dy, stride, pad, dil):
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad, dil)
    y.backward(dy)
    return x.grad, w.grad


@pytest.mark.parametrize("x3", [False, True], ids=["f32", "bf16x3"])
@pytest.mark.parametrize("case", CASES)
def test_conv_wgrad_matches_autograd(case, x3):
    """tt_conv2d_wgrad (exact f32 products) and tt_conv2d_wgrad_x3 (the forward's bf16x3 arithmetic: layers with >= 128
    channels on both sides take the LDS-staged bf16-MFMA kernel, the rest falls through to the f32 kernels)."""
    from thinktwice_amd import ops, weights
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(Cin * 7 + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5
    OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    dy = torch.randn(N, Cout, OH, OW, generator=g)
    _, dw_ref = _ref(x, w, dy, stride, pad, dil)
    cp = (Cin + 3) // 4 * 4
    xq = weights.to_channel_last(x, torch.float32).cuda()                 # (N, H, W, cp), zero-padded channels
    dyq = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dw = ops.conv2d_wgrad(xq, dyq, k, k, stride, pad, dil, cin=Cin, cin_pad=cp, x3=x3)
    torch.cuda.synchronize()
    got = dw[..., :Cin].permute(0, 3, 1, 2).cpu()
    err = float((got - dw_ref).abs().max() / dw_ref.abs().max())
    assert err < (1e-4 if x3 else 2e-5), err
    assert float(dw[..., Cin:].abs().max()) == 0.0 if cp > Cin else True
    # accumulate into an existing gradient, bit-reproducible
    dw2 = ops.conv2d_wgrad(xq, dyq, k, k, stride, pad, dil, cin=Cin, cin_pad=cp, out=dw.clone(), accumulate=True, x3=x3)
    assert float((dw2 - 2 * dw).abs().max()) <= 1e-5 * float(dw.abs().max())       # (g + partials) vs 2 * sum: rounding only
    assert torch.equal(ops.conv2d_wgrad(xq, dyq, k, k, stride, pad, dil, cin=Cin, cin_pad=cp, x3=x3), dw)


@pytest.mark.parametrize("case", [c for c in CASES if c[3] >= 32])
@pytest.mark.parametrize("x3", [False, True])
def test_conv_dgrad_matches_autograd(case, x3):
    from thinktwice_amd import ops, weights
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(Cin * 3 + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5
    OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    dy = torch.randn(N, Cout, OH, OW, generator=g)
    dx_ref, _ = _ref(x, w, dy, stride, pad, dil)
    wq = w.permute(0, 2, 3, 1).contiguous().cuda()                       # [Cout][KH][KW][Cin]
    dyq = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dx = ops.conv2d_dgrad(dyq, wq, (H, W), stride, pad, dil, x3=x3)
    torch.cuda.synchronize()
    got = dx.permute(0, 3, 1, 2).cpu()
    assert got.shape == dx_ref.shape
    err = float((got - dx_ref).abs().max() / dx_ref.abs().max())
    assert err < (2e-4 if x3 else 2e-5), err


@pytest.mark.parametrize("act,nres,C", [(1, 0, 64), (1, 1, 256), (1, 2, 96), (0, 0, 320), (2, 0, 32), (0, 1, 1280)])
def test_conv_epilogue_bwd_matches_autograd(act, nres, C):
    """y = act(scale*conv + shift + res1 + res2): gradients w.r.t. conv, the residuals, scale and shift from dy and the
    saved y (tt_conv_epilogue_bwd) against autograd; then a whole conv + folded-BN + ReLU + residual layer backward
    (epilogue -> wgrad / dgrad) against autograd of the same layer."""
    from thinktwice_amd import ops
    g = torch.Generator().manual_seed(act * 100 + nres * 10 + C)
    M = 1531
    conv = torch.randn(M, C, generator=g, requires_grad=True)
    scale = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    shift = (torch.randn(C, generator=g) * 0.3).requires_grad_(True)
    res = [torch.randn(M, C, generator=g, requires_grad=True) for _ in range(nres)]
    pre = conv * scale + shift + sum(res)
    y = {0: pre, 1: torch.relu(pre), 2: torch.sigmoid(pre)}[act]
    dy = torch.randn(M, C, generator=g)
    y.backward(dy)
    r = [t.detach().cuda() for t in res] + [None, None]
    dconv, dres, dscale, dshift = ops.conv_epilogue_bwd(dy.cuda(), y.detach().cuda(), scale.detach().cuda(),
                                                        shift.detach().cuda(), act, r[0], r[1], want_dres=nres > 0)
    torch.cuda.synchronize()
    tol = 2e-4 if act == 2 else 2e-5          # sigmoid recovers the pre-activation through logit(y)

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max())
    assert rel(dconv, conv.grad) < tol
    assert rel(dshift, shift.grad) < tol and rel(dscale, scale.grad) < 5 * tol
    if nres:
        assert rel(dres, res[0].grad) < tol


def test_conv_bn_relu_residual_layer_backward():
    from thinktwice_amd import ops
    g = torch.Generator().manual_seed(11)
    N, H, W, Cin, Cout = 2, 24, 32, 64, 128
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5).requires_grad_(True)
    gamma = (torch.rand(Cout, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(Cout, generator=g) * 0.2).requires_grad_(True)
    mean, var = torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5
    res = torch.randn(N, Cout, H, W, generator=g, requires_grad=True)
    y = torch.relu(F.batch_norm(F.conv2d(x, w, None, 1, 1), mean, var, gamma, beta, False, 0.0, 1e-5) + res)
    dy = torch.randn(N, Cout, H, W, generator=g)
    y.backward(dy)
    sigma = torch.sqrt(var + 1e-5)
    scale, shift = (gamma / sigma).detach(), (beta - mean * gamma / sigma).detach()
    cl = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()   # noqa: E731
    xq, wq, yq, dyq, rq = cl(x), cl(w), cl(y), cl(dy), cl(res)
    dconv, dres, dscale, dshift = ops.conv_epilogue_bwd(dyq, yq, scale.cuda(), shift.cuda(), 1, rq, want_dres=True)
    dconv = dconv.view(N, H, W, Cout)
    dw = ops.conv2d_wgrad(xq, dconv, 3, 3, 1, 1, 1)
    dx = ops.conv2d_dgrad(dconv, wq, (H, W), 1, 1, 1, x3=True)
    dgamma = (dscale.cpu() - mean * dshift.cpu()) / sigma
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max())
    assert rel(dw.permute(0, 3, 1, 2), w.grad) < 3e-5
    assert rel(dx.permute(0, 3, 1, 2), x.grad) < 2e-4
    assert rel(dres.view(N, H, W, Cout).permute(0, 3, 1, 2), res.grad) < 1e-6
    assert rel(dgamma, gamma.grad) < 1e-4 and rel(dshift, beta.grad) < 3e-5


@pytest.mark.parametrize("Cin,Cout", [(16, 16), (32, 64), (64, 64), (128, 128), (48, 160), (130, 24)])
def test_gather_conv_wgrad_matches_dense_sum(Cin, Cout):
    """tt_gather_conv_wgrad: dW[co][t][ci] = sum over the LIVE rows m of dy[m][co] * x[nbr[m][t]][ci] (nbr = -1: no input) for
    every wave-tile form of the kernel (32 / 64 / 128 channels a side), a ragged live-row count and empty taps."""
    from thinktwice_amd import ops
    g = torch.Generator().manual_seed(Cin * 31 + Cout)
    R_in, M, live, taps = 700, 1500, 1237, 27
    x = torch.randn(R_in, Cin, generator=g)
    dy = torch.randn(M, Cout, generator=g)
    nbr = torch.randint(0, R_in, (M, taps), generator=g, dtype=torch.int32)
    nbr[torch.rand(M, taps, generator=g) < 0.6] = -1
    nbr[:, 5] = -1                                            # a tap no row uses
    xg = torch.where((nbr >= 0).unsqueeze(-1), x[nbr.clamp_min(0).long()], torch.zeros(()))   # (M, taps, Cin)
    ref = torch.einsum("mo,mti->oti", dy[:live].double(), xg[:live].double()).float()
    cin_pad = (Cin + 3) // 4 * 4
    m_dev = torch.tensor([live], dtype=torch.int32, device="cuda")
    got = ops.gather_conv_wgrad(x.cuda(), nbr.cuda(), m_dev, dy.cuda(), taps, cin_pad=cin_pad).cpu()
    assert got.shape == (Cout, 1, taps, cin_pad)
    assert float(got[..., Cin:].abs().max()) == 0.0 if cin_pad > Cin else True
    err = float((got[:, 0, :, :Cin] - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
    again = ops.gather_conv_wgrad(x.cuda(), nbr.cuda(), m_dev, dy.cuda(), taps, cin_pad=cin_pad).cpu()
    assert torch.equal(got, again)                            # ordered partial sums: bit-reproducible


def test_frozen_bn_layer_backward_with_zero_and_tiny_gamma_channels():
    """a folded BatchNorm channel with gamma == 0 (mmdet zero_init_residual, pruned channels) or gamma ~ 1e-8
    must not turn dgamma into NaN / garbage.  The taped layer recomputes the raw convolution for such layers
    (autodiff.refresh_small_scale_flags) and the kernel sums g * conv directly; without the flag a zero scale contributes
    0 instead of 0/0."""
    from thinktwice_amd import autodiff, layers, ops, weights
    g = torch.Generator().manual_seed(21)
    N, H, W, Cin, Cout = 2, 12, 16, 32, 64
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5).requires_grad_(True)
    gamma = torch.rand(Cout, generator=g) + 0.5
    gamma[3], gamma[10], gamma[40] = 0.0, 1e-8, -1e-9
    gamma.requires_grad_(True)
    beta = (torch.randn(Cout, generator=g) * 0.2).requires_grad_(True)
    mean, var = torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5
    y = torch.relu(F.batch_norm(F.conv2d(x, w, None, 1, 1), mean, var, gamma, beta, False, 0.0, 1e-5))
    R = torch.randn(N, Cout, H, W, generator=g)
    (y * R).sum().backward()
    sd = {"l.weight": w.detach(), "l_bn.weight": gamma.detach(), "l_bn.bias": beta.detach(), "l_bn.running_mean": mean,
          "l_bn.running_var": var}
    autodiff.clear_metas()
    conv = layers.conv_from_sd(sd, "l", torch.float32, torch.device("cuda"), bn="l_bn", pad=1, act="relu")
    assert autodiff.refresh_small_scale_flags() == 1
    xq = weights.to_channel_last(x, torch.float32).cuda()
    for flagged in (True, False):
        autodiff.CONV_META[conv.w].small_scale = flagged
        with autodiff.Tape(x3=False) as tape:
            out = conv(xq)
            tape.seed(out, R.permute(0, 2, 3, 1))
            tape.backward()
        torch.cuda.synchronize()
        dg, db = tape.param_grads["l_bn.weight"].cpu(), tape.param_grads["l_bn.bias"].cpu()
        assert torch.isfinite(dg).all() and torch.isfinite(db).all() and torch.isfinite(tape.param_grads["l.weight"]).all()
        assert float((db - beta.grad).abs().max()) < 3e-5 * float(beta.grad.abs().max())
        if flagged:         # exact everywhere, the degenerate channels included
            assert float((dg - gamma.grad).abs().max()) < 1e-4 * float(gamma.grad.abs().max())
        else:               # ordinary channels exact, degenerate ones finite (their information is not in the saved output)
            ok = torch.ones(Cout, dtype=torch.bool)
            ok[[3, 10, 40]] = False
            assert float((dg - gamma.grad)[ok].abs().max()) < 1e-4 * float(gamma.grad.abs().max())
    autodiff.clear_metas()
