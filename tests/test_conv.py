"""Implicit-GEMM conv / linear kernel (tt_conv2d_fwd) vs a plain PyTorch fp32 reference of the
same op on CPU.  Tolerances: f32 mode 1e-4 relative-to-max (exact-f32 MFMA, different sum order);
bf16 mode 2e-2 (inputs rounded to bf16; reference evaluated on the same rounded inputs)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mk(shape, g, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


def _run(dt, N, H, W, Cin, Cout, k, stride=1, pad=0, dil=1, act=0, use_bn=True, res=0, seed=0):
    from thinktwice_amd import ops, weights
    g = torch.Generator().manual_seed(seed)
    x = _mk((N, Cin, H, W), g)
    w = _mk((Cout, Cin, k, k), g, (Cin * k * k) ** -0.5)
    scale = (torch.rand(Cout, generator=g) + 0.5) if use_bn else None
    shift = _mk((Cout,), g, 0.3)
    xq = weights.to_channel_last(x, dt).cuda()
    wq = weights.prep_conv_weight(w, dt).cuda()
    x_ref = xq.float().cpu()[..., :Cin].permute(0, 3, 1, 2)
    w_ref = wq.float().cpu()[..., :Cin].permute(0, 3, 1, 2)
    ref = F.conv2d(x_ref, w_ref, None, stride, pad, dil)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    ref = ref + shift.view(1, -1, 1, 1)
    r1 = r2 = None
    if res >= 1:
        r1 = _mk(tuple(ref.shape), g).permute(0, 2, 3, 1).contiguous().to(dt)
        ref = ref + r1.float().permute(0, 3, 1, 2)
    if res >= 2:
        r2 = _mk(tuple(ref.shape), g).permute(0, 2, 3, 1).contiguous().to(dt)
        ref = ref + r2.float().permute(0, 3, 1, 2)
    ref = {0: lambda t: t, 1: F.relu, 2: torch.sigmoid, 3: F.gelu, 4: F.softplus}[act](ref)
    out = ops.conv2d(xq, wq, stride=stride, pad=pad, dil=dil,
                     scale=None if scale is None else scale.cuda(), shift=shift.cuda(), act=act,
                     res1=None if r1 is None else r1.cuda(), res2=None if r2 is None else r2.cuda())
    got = out.float().cpu().permute(0, 3, 1, 2)
    tol = 1e-4 if dt == torch.float32 else (3e-3 if dt == torch.float16 else 2e-2)
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    assert got.shape == ref.shape
    assert err < tol, err


CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, act, bn, res
    (2, 13, 17, 64, 128, 3, 1, 1, 1, 1, True, 1),     # ragged M, BasicBlock-like epilogue
    (1, 28, 56, 256, 80, 1, 1, 0, 1, 0, False, 0),    # 1x1, Cout tail in a 128 tile
    (2, 40, 48, 3, 64, 7, 2, 3, 1, 1, True, 0),       # ResNet stem (Cin 3 -> padded)
    (1, 21, 21, 512, 256, 3, 2, 1, 1, 1, True, 0),    # stride 2
    (1, 28, 56, 64, 64, 3, 1, 6, 6, 1, True, 0),      # ASPP dilation 6
    (1, 30, 30, 64, 64, 3, 1, 18, 18, 1, True, 0),    # ASPP dilation 18
    (2, 16, 16, 64, 12, 1, 1, 0, 1, 0, False, 0),     # seg head Cout=12
    (2, 9, 9, 512, 18, 3, 1, 1, 1, 0, False, 0),      # DCN offset conv Cout=18
    (3, 21, 21, 38, 32, 3, 1, 1, 1, 1, False, 0),     # SpatialGRU conv, Cin=38 (padded)
    (2, 21, 21, 256, 256, 3, 1, 1, 1, 1, True, 2),    # conv_fusion: two residuals
    (5, 1, 1, 1543, 512, 1, 1, 0, 1, 3, False, 0),    # linear 1543->512 + GELU, 5 rows
    (300, 1, 1, 256, 1024, 1, 1, 0, 1, 2, False, 0),  # linear + sigmoid
    (1, 7, 7, 128, 256, 3, 1, 0, 1, 4, False, 0),     # valid conv, softplus
    # M >= 2048 with Cin % BK == 0: the LDS-DMA (glds) 3-stage kernel
    (2, 40, 48, 64, 128, 3, 1, 1, 1, 1, True, 1),     # 256x128 tile, ragged M (3840), residual
    (1, 64, 64, 128, 64, 1, 1, 0, 1, 0, False, 0),    # 256x64 tile, 1x1
    (2, 56, 100, 256, 512, 3, 2, 1, 1, 1, True, 0),   # stride 2, many N tiles
    (2, 40, 48, 64, 64, 3, 1, 6, 6, 1, True, 2),      # dilation 6, two residuals
    (3, 30, 30, 192, 200, 3, 1, 1, 1, 3, False, 0),   # Cout tail (200), gelu
    (1, 50, 60, 64, 256, 1, 1, 0, 1, 2, True, 1),     # single K tile (bf16), sigmoid
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_conv_matches_torch(dt, case):
    N, H, W, Cin, Cout, k, s, p, d, act, bn, res = case
    _run(dt, N, H, W, Cin, Cout, k, s, p, d, act, bn, res, seed=Cin + Cout + k)


X3_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, act, bn, res   (M > 4096, Cin % 32 == 0: the bf16x3 LDS-DMA kernel)
    (3, 40, 48, 64, 128, 3, 1, 1, 1, 1, True, 1),     # 256x128 tile (8 waves x 32x128), ragged M (5760), residual + ReLU
    (2, 64, 64, 128, 64, 1, 1, 0, 1, 0, False, 0),    # 256x64 tile, 1x1
    (2, 56, 100, 256, 512, 3, 2, 1, 1, 1, True, 0),   # 256x256 tile (8 waves x 64x128), stride 2
    (3, 40, 48, 64, 64, 3, 1, 6, 6, 1, True, 2),      # dilation 6, two residuals (rolled epilogue)
    (5, 30, 30, 192, 200, 3, 1, 1, 1, 3, False, 0),   # Cout tail (200) in the 128 tile, GELU
    (2, 50, 60, 64, 256, 1, 1, 0, 1, 2, True, 1),     # 2 K tiles only, sigmoid
    (2, 36, 64, 256, 1280, 1, 1, 0, 1, 0, False, 0),  # value_proj-like N = 1280
    (2, 21, 21, 512, 256, 3, 1, 1, 1, 1, True, 0),    # M = 882 <= 4096: the exact f32 small-M kernel on `w`
    (3, 300, 300, 64, 12, 3, 1, 1, 1, 0, False, 0),   # 256x32 tile: few output channels over >= 2^18 rows (the fused seg head)
]


@pytest.mark.parametrize("case", X3_CASES)
def test_conv_bf16x3_matches_torch_f32(case):
    """Precision mode "f32x3" (f32 storage, three bf16 MFMAs per product on pre-split weights): within 1e-4 of the f32
    reference -- two orders of magnitude tighter than one bf16 MFMA -- on the shapes the LDS-DMA kernel takes, and
    exactly the f32 path elsewhere."""
    from thinktwice_amd import ops, weights
    N, H, W, Cin, Cout, k, stride, pad, dil, act, use_bn, res = case
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = _mk((N, Cin, H, W), g)
    w = _mk((Cout, Cin, k, k), g, (Cin * k * k) ** -0.5)
    scale = (torch.rand(Cout, generator=g) + 0.5) if use_bn else None
    shift = _mk((Cout,), g, 0.3)
    xq = weights.to_channel_last(x, torch.float32).cuda()
    wq = weights.prep_conv_weight(w, torch.float32).cuda()
    wx = weights.split_pairs_x3(wq)
    assert wx is not None and wx.shape == wq.shape
    ref = F.conv2d(x, w, None, stride, pad, dil)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    ref = ref + shift.view(1, -1, 1, 1)
    r1 = r2 = None
    if res >= 1:
        r1 = _mk(tuple(ref.shape), g).permute(0, 2, 3, 1).contiguous()
        ref = ref + r1.permute(0, 3, 1, 2)
    if res >= 2:
        r2 = _mk(tuple(ref.shape), g).permute(0, 2, 3, 1).contiguous()
        ref = ref + r2.permute(0, 3, 1, 2)
    ref = {0: lambda t: t, 1: F.relu, 2: torch.sigmoid, 3: F.gelu, 4: F.softplus}[act](ref)
    out = ops.conv2d(xq, wq, stride=stride, pad=pad, dil=dil, scale=None if scale is None else scale.cuda(),
                     shift=shift.cuda(), act=act, res1=None if r1 is None else r1.cuda(),
                     res2=None if r2 is None else r2.cuda(), w_x3=wx)
    got = out.cpu().permute(0, 3, 1, 2)
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    assert got.shape == ref.shape
    assert err < 1e-4, err
    # and it is not the exact-f32 kernel in disguise: the bf16x3 result differs from the f32 one in the low bits
    if N * got.shape[2] * got.shape[3] > 4096:        # (M <= 4096 runs the latency kernel conv_small.hip in exact f32)
        exact = ops.conv2d(xq, wq, stride=stride, pad=pad, dil=dil, scale=None if scale is None else scale.cuda(),
                           shift=shift.cuda(), act=act, res1=None if r1 is None else r1.cuda(),
                           res2=None if r2 is None else r2.cuda())
        assert not torch.equal(exact, out)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_deconv2x2_pixel_shuffle(dt):
    from thinktwice_amd import ops, weights
    g = torch.Generator().manual_seed(3)
    x = _mk((2, 256, 7, 14), g)
    w = _mk((256, 128, 2, 2), g, 256 ** -0.5)
    b = _mk((128,), g, 0.2)
    xq = weights.to_channel_last(x, dt).cuda()
    wq = weights.prep_deconv2x2_weight(w, dt).cuda()
    ref = F.conv_transpose2d(xq.float().cpu().permute(0, 3, 1, 2),
                             w.to(dt).float(), b, stride=2)
    # write into the first 128 channels of a 384-channel concat buffer (UnetLayer.forward: cat)
    buf = torch.zeros(2, 14, 28, 384, dtype=dt).cuda()
    ops.conv2d(xq, wq, shift=b.cuda(), pixel_shuffle2=True, out=buf, out_coff=0)
    got = buf.float().cpu()[..., :128].permute(0, 3, 1, 2)
    tol = 1e-4 if dt == torch.float32 else (3e-3 if dt == torch.float16 else 2e-2)
    assert float((got - ref).abs().max() / ref.abs().max()) < tol
    assert float(buf[..., 128:].abs().max()) == 0.0


def test_channel_offsets_strided_output_and_per_image_shift():
    from thinktwice_amd import ops, weights
    dt = torch.float32
    g = torch.Generator().manual_seed(4)
    x = _mk((8, 96, 6, 5), g)                      # use channels [32, 96) only
    w = _mk((256, 64, 1, 1), g, 0.1)
    b = _mk((256,), g, 0.2)
    cam = _mk((4, 256), g)                          # per-camera embedding (DEC:392), image n -> cam n%4
    xq = weights.to_channel_last(x, dt).cuda()
    wq = weights.prep_conv_weight(w, dt).cuda()
    ref = F.conv2d(x[:, 32:], w, b) + cam[torch.arange(8) % 4][:, :, None, None]
    # value-buffer layout: [image][total_positions=50][256], this level occupies rows [20, 50)
    val = torch.zeros(8, 50, 256).cuda()
    flat = val.view(-1)
    out_view = torch.as_strided(flat, (8, 6, 5, 256), (50 * 256, 5 * 256, 256, 1), 20 * 256)
    import ctypes
    d = ops._ConvDesc()
    d.in_ = xq.data_ptr(); d.N = 8; d.H = 6; d.W = 5; d.Cin = 64; d.in_cstride = 96; d.in_coff = 32
    d.weight = wq.data_ptr(); d.Cout = 256; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0; d.dil = 1
    d.out = out_view.data_ptr(); d.OH = 6; d.OW = 5; d.out_cstride = 256; d.out_coff = 0
    d.out_nstride = 50 * 256
    bc, cc = b.cuda(), cam.cuda().contiguous()
    d.shift = bc.data_ptr(); d.shift_n = cc.data_ptr(); d.shift_n_mod = 4
    d.act = 0; d.dtype = 0; d.out_dtype = 0
    ops.check(ops.lib().tt_conv2d_fwd(ctypes.byref(d), ops.cur_stream(xq.device)), "conv")
    got = out_view.cpu().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-4
    assert float(val[:, :20].abs().max()) == 0.0


def test_conv_rejects_unpadded_channels():
    from thinktwice_amd import _lib, ops
    x = torch.zeros(1, 4, 4, 6).cuda()
    w = torch.zeros(8, 1, 1, 6).cuda()
    with pytest.raises(_lib.TTError):
        ops.conv2d(x, w)


def test_splitk_workspace_path_matches_torch():
    """tt_conv_desc.splitk_ws: K split over gridDim.y + finalize kernel (kept for callers that want it)."""
    from thinktwice_amd import ops, weights
    g = torch.Generator().manual_seed(11)
    x = _mk((2, 64, 21, 21), g)
    w = _mk((128, 64, 3, 3), g, (64 * 9) ** -0.5)
    b = _mk((128,), g, 0.1)
    xq = weights.to_channel_last(x).cuda()
    wq = weights.prep_conv_weight(w, torch.float32).cuda()
    ws = torch.zeros(2 * 21 * 21, 128, device="cuda")
    out = ops.conv2d(xq, wq, pad=1, shift=b.cuda(), act=1, splitk_ws=ws)
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    got = out.cpu().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-4


@pytest.mark.parametrize("dt,Cin,Cout,rows_in,M,live", [
    (torch.bfloat16, 64, 64, 5000, 6000, 5300),      # LDS-DMA gather kernel, 256x64 tile
    (torch.bfloat16, 128, 128, 3000, 4096, 4096),    # LDS-DMA gather kernel, 256x128 tile, 2 K tiles per tap
    (torch.bfloat16, 32, 32, 3000, 4000, 3500),      # LDS-DMA gather kernel, 2 taps per 128 B row, 256x32 tile
    (torch.bfloat16, 16, 16, 3000, 2500, 2400),      # 4 taps per 128 B row, last K tile ragged (27 taps)
    (torch.bfloat16, 32, 64, 1000, 1500, 1400),      # M < 2048: register-staged gather kernel
    (torch.float32, 16, 32, 2000, 3000, 2999),
    (torch.float16, 64, 64, 5000, 6000, 5300),
    (torch.float16, 16, 16, 3000, 2500, 2400),
    ("f32x3", 64, 64, 5000, 6000, 5300),             # bf16x3 gather: 256x64 tile, 2 K tiles per tap
    ("f32x3", 128, 128, 3000, 4096, 4096),           # 256x128 tile, 4 K tiles per tap
    ("f32x3", 32, 32, 3000, 4000, 3500),             # one tap per 128 B row, 256x32 tile
    ("f32x3", 16, 16, 3000, 2500, 2400),             # 2 taps per 128 B row, last K tile ragged (27 taps)
])
def test_gathered_sparse_conv_matches_torch(dt, Cin, Cout, rows_in, M, live):
    """Sparse conv as a gathered GEMM (tt_conv2d_fwd gather mode): rulebook [M, 27] with -1 holes, device-side live
    row count; rows >= live are not written.  Reference: explicit gather + matmul in f32 on the rounded inputs."""
    from thinktwice_amd import ops, weights
    x3 = dt == "f32x3"
    dt = torch.float32 if x3 else dt
    g = torch.Generator().manual_seed(3)
    taps = 27
    feats = (torch.randn(rows_in, Cin, generator=g)).to(dt)
    w = (torch.randn(Cout, 1, taps, Cin, generator=g) * (taps * Cin) ** -0.5).to(dt)
    nbr = torch.randint(0, rows_in, (M, taps), generator=g, dtype=torch.int32)
    nbr[torch.rand(M, taps, generator=g) < 0.4] = -1
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.3
    res = torch.randn(M, Cout, generator=g).to(dt)
    m_dev = torch.tensor([live], dtype=torch.int32)
    wd = w.cuda()
    out = ops.gather_conv(feats.cuda(), nbr.cuda(), m_dev.cuda(), wd, scale=scale.cuda(), shift=shift.cuda(),
                          act=1, res=res.cuda(), w_x3=weights.split_pairs_x3(wd) if x3 else None)
    torch.cuda.synchronize()
    f32 = feats.float()
    gathered = torch.where((nbr >= 0).unsqueeze(-1), f32[nbr.clamp_min(0).long()], torch.zeros(()))   # [M, taps, Cin]
    ref = torch.relu(gathered.reshape(M, -1) @ w.float().reshape(Cout, -1).t() * scale + shift + res.float())
    got = out.float().cpu()
    tol = 1e-4 if dt == torch.float32 else (3e-3 if dt == torch.float16 else 2e-2)
    err = float((got[:live] - ref[:live]).abs().max() / ref[:live].abs().max())
    assert err < tol, err


@pytest.mark.parametrize("dt,Cin,Cout", [(torch.bfloat16, 16, 16), (torch.bfloat16, 32, 64), (torch.bfloat16, 64, 64),
                                         (torch.bfloat16, 128, 128), (torch.float16, 32, 32), ("f32x3", 16, 32),
                                         ("f32x3", 64, 64), ("f32x3", 128, 128)])
def test_planned_sparse_conv_skips_empty_taps_and_matches_torch(dt, Cin, Cout):
    """Mask-sorted gathered GEMM (tt_sp_tile_plan + tt_conv2d_fwd row_perm / row_mask): rows carry 1-3 of the 27 taps
    (what a LiDAR rulebook looks like), the plan sorts them by mask and every 256-row tile walks only the union of its
    taps.  Same result as the unplanned launch and as an explicit gather + matmul; rows >= live are not written."""
    from thinktwice_amd import ops, weights
    x3 = dt == "f32x3"
    dt = torch.float32 if x3 else dt
    g = torch.Generator().manual_seed(Cin + Cout)
    taps, rows_in, M, live = 27, 4000, 7000, 6500
    feats = torch.randn(rows_in, Cin, generator=g).to(dt)
    w = (torch.randn(Cout, 1, taps, Cin, generator=g) * (3 * Cin) ** -0.5).to(dt)
    nbr = torch.full((M, taps), -1, dtype=torch.int32)
    for _ in range(3):      # up to 3 taps per row, the centre tap for most rows
        t = torch.randint(0, taps, (M,), generator=g)
        t[torch.rand(M, generator=g) < 0.5] = 13
        nbr[torch.arange(M), t] = torch.randint(0, rows_in, (M,), generator=g, dtype=torch.int32)
    nbr[:5] = -1            # rows without any tap: pure bias
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.3
    res = torch.randn(M, Cout, generator=g).to(dt)
    m_dev = torch.tensor([live], dtype=torch.int32).cuda()
    wd, nd, fd = w.cuda(), nbr.cuda(), feats.cuda()
    wx = weights.split_pairs_x3(wd) if x3 else None
    plan = ops.sp_tile_plan(nd, m_dev)
    # the plan is a permutation of the live rows, sorted by mask
    perm = plan[0].cpu()[:live].long()
    assert torch.equal(torch.sort(perm).values, torch.arange(live))
    masks = ((nbr[:live] >= 0).long() << torch.arange(taps)).sum(1)
    assert torch.equal(plan[1].cpu()[:live].long() & 0xFFFFFFFF, masks[perm])
    assert bool((masks[perm][1:] >= masks[perm][:-1]).all())
    kw = dict(scale=scale.cuda(), shift=shift.cuda(), act=1, res=res.cuda(), w_x3=wx)
    sentinel = 123.0
    outs = []
    for pl in (plan, None):
        pre = torch.full((M, Cout), sentinel, dtype=dt, device="cuda")
        out = ops.gather_conv(fd, nd, m_dev, wd, plan=pl, **kw)
        outs.append(out.float().cpu())
    torch.cuda.synchronize()
    f32 = feats.float()
    gathered = torch.where((nbr >= 0).unsqueeze(-1), f32[nbr.clamp_min(0).long()], torch.zeros(()))
    ref = torch.relu(gathered.reshape(M, -1) @ w.float().reshape(Cout, -1).t() * scale + shift + res.float())
    tol = 1e-4 if dt == torch.float32 else (3e-3 if dt == torch.float16 else 2e-2)
    for got in outs:
        err = float((got[:live] - ref[:live]).abs().max() / ref[:live].abs().max())
        assert err < tol, err


def test_row_run_stem_matches_conv7x7s2():
    """bf16 7x7/2 stem as a KH=7, KW=1, Cin=64 'row-run' convolution over a zero-bordered 8-channel image
    (lss.py _ResNet50.stem_rr) == F.conv2d(x, w, stride=2, padding=3) on the same rounded operands."""
    from thinktwice_amd import ops
    g = torch.Generator().manual_seed(5)
    N, H, W = 6, 64, 96
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * (3 * 49) ** -0.5
    shift = torch.randn(64, generator=g) * 0.1
    dt = torch.bfloat16
    xp = torch.zeros(N, H + 6, W + 8, 8, dtype=dt, device="cuda")
    ops.nchw_to_nhwc_border(x.cuda(), xp, 3, 3)
    wr = torch.zeros(64, 7, 1, 64, dtype=dt)
    wr[:, :, 0, :56] = F.pad(w.permute(0, 2, 3, 1), (0, 5)).reshape(64, 7, 56).to(dt)
    out = ops.conv2d(xp, wr.cuda(), stride=2, pad=0, shift=shift.cuda(), act=1, in_cstride=8,
                     out_hw=(H // 2, W // 2))
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(x.to(dt).float(), w.to(dt).float(), None, 2, 3) + shift.view(1, -1, 1, 1))
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 2e-2, err
    # the border must still be zero (the conversion writes the interior only)
    assert float(xp[:, :3].abs().max()) == 0 and float(xp[:, :, :3].abs().max()) == 0 and float(xp[:, :, W + 3:].abs().max()) == 0


def _grid_rulebook(B, D, H, W, occ, stride, g):
    """Rulebook of a 3x3x3 sparse conv on a random occupancy grid, rows in (b, z, y, x) cell order like
    csrc/lidar.hip builds them.  stride 1: SubM (outputs = inputs, pad 1); stride 2: spconv SparseConv3d pad 1."""
    vol = torch.rand(B, D, H, W, generator=g) < occ
    idx = torch.full((B, D, H, W), -1, dtype=torch.int64)
    idx[vol] = torch.arange(int(vol.sum()))
    rows_in = int(vol.sum())
    pad = F.pad(idx, (1, 1, 1, 1, 1, 1), value=-1)
    if stride == 1:
        oc = vol.nonzero()
    else:
        od = [(n + 2 - 3) // 2 + 1 for n in (D, H, W)]
        hit = F.max_pool3d(F.pad(vol.float()[:, None], (1, 1, 1, 1, 1, 1)), 3, 2)[:, 0, :od[0], :od[1], :od[2]] > 0
        oc = hit.nonzero()
    b, z, y, x = oc.unbind(1)
    taps = []
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                taps.append(pad[b, z * stride + kz, y * stride + ky, x * stride + kx])
    return torch.stack(taps, 1).to(torch.int32), rows_in


@pytest.mark.parametrize("Cin,Cout,stride,occ,dims", [
    (32, 32, 1, 0.10, (2, 10, 60, 60)),      # level-2 like: sparse lines
    (64, 64, 1, 0.65, (2, 6, 40, 44)),       # level-3 like: near-dense
    (128, 128, 1, 0.85, (3, 3, 30, 34)),     # level-4 like
    (32, 64, 2, 0.10, (2, 11, 60, 60)),      # strided: every other input line
    (64, 128, 2, 0.65, (2, 5, 44, 40)),
    (64, 64, 1, 0.002, (1, 9, 420, 420)),    # isolated sites: most groups empty, centre tap only
])
def test_run_staged_sparse_conv_on_cell_ordered_rulebooks(Cin, Cout, stride, occ, dims):
    """csrc/sp_conv_runs.hip (bf16x3, 27 taps, >= 32 channels): rulebooks with the locality of real levels (rows in cell
    order), a live-row count below the allocation, garbage beyond it.  Reference: explicit gather + matmul in f32."""
    from thinktwice_amd import ops, weights
    g = torch.Generator().manual_seed(Cin + Cout + stride)
    nbr, rows_in = _grid_rulebook(*dims, occ, stride, g)
    live = nbr.shape[0]
    M = live + 300
    nbr_alloc = torch.randint(-5, 10 ** 6, (M, 27), generator=g, dtype=torch.int32)   # rows >= live: garbage
    nbr_alloc[:live] = nbr
    feats = torch.randn(rows_in, Cin, generator=g)
    w = torch.randn(Cout, 1, 27, Cin, generator=g) * (27 * Cin) ** -0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.3
    res = torch.randn(M, Cout, generator=g)
    wd = w.cuda()
    out = torch.full((M, Cout), 123.0, device="cuda")
    ops.gather_conv(feats.cuda(), nbr_alloc.cuda(), torch.tensor([live], dtype=torch.int32).cuda(), wd,
                    scale=scale.cuda(), shift=shift.cuda(), act=1, res=res.cuda(), w_x3=weights.split_pairs_x3(wd), out=out)
    torch.cuda.synchronize()
    if live >= 2048:
        assert ops._last_conv_kernel().startswith("sp_conv_runs_kernel"), ops._last_conv_kernel()
    gathered = torch.where((nbr >= 0).unsqueeze(-1), feats[nbr.clamp_min(0).long()], torch.zeros(()))
    ref = torch.relu(gathered.reshape(live, -1) @ w.reshape(Cout, -1).t() * scale + shift + res[:live])
    got = out.cpu()
    err = float((got[:live] - ref).abs().max() / ref.abs().max())
    assert err < 1e-4, err
    assert bool((got[live:] == 123.0).all())          # rows beyond the live count are not written


# Shapes that take the round-4 hand-pipelined bf16x3 kernels (csrc/conv_x3_pipe.hip): K >= 1152, Cout % 128 == 0, M > 4096.
# (N, H, W, Cin, Cout, k, stride, act, bn, residuals, in_window, expected kernel)
PIPE_CASES = [   # (M sized so that the tile cost model picks the wide tile: ~256 tiles of 256 rows)
    (4, 128, 128, 128, 256, 3, 1, 1, True, 1, False, "conv_x3_run3_kernel<256>"),    # 3x3 "same": run-staged, 256-wide
    (5, 100, 130, 160, 256, 3, 1, 0, False, 0, False, "conv_x3_run3_kernel<256>"),   # ragged M (253.9 tiles), 5 channel chunks
    (2, 128, 128, 128, 512, 3, 1, 1, True, 0, False, "conv_x3_run3_kernel<256>"),    # two column tiles
    (4, 128, 128, 128, 128, 3, 1, 1, True, 2, False, "conv_x3_run3_kernel<128>"),    # 128-wide, two residuals (rolled epilogue)
    (4, 128, 128, 128, 256, 3, 1, 3, False, 0, True, "conv_x3_run3_kernel<256>"),    # input = channel window of a wider buffer, GELU
    (4, 256, 256, 128, 256, 3, 2, 1, True, 0, False, "conv_x3_pipe_kernel<4, 1>"),   # stride 2: per-tap pipelined tile
    (4, 128, 128, 1152, 256, 1, 1, 1, True, 1, False, "conv_x3_pipe_kernel<4, 1>"),  # 1x1 with a long K
    (4, 256, 256, 128, 128, 3, 2, 0, False, 0, False, "conv_x3_pipe_kernel<4, 1, 128>"),
    (1, 1, 65536, 128, 256, 3, 1, 1, True, 0, False, "conv_x3_run3_kernel<256>"),    # ONE image row: no vertical neighbours at all
    (64, 32, 32, 128, 256, 3, 1, 1, False, 0, False, "conv_x3_run3_kernel<256>"),    # 32 x 32 images: 4 images per tile, 8 rows per image row
]


@pytest.mark.parametrize("case", PIPE_CASES, ids=[c[-1] + f"-{i}" for i, c in enumerate(PIPE_CASES)])
def test_pipelined_bf16x3_tiles_match_torch_f32(case, monkeypatch):
    """The 4-wave hand-pipelined tiles and their run-staged 3x3 form against torch f32 (1e-4) and against the exact-f32
    kernel, on shapes with padding on every side, ragged M, image-row boundaries inside a tile, stride 2 and channel-window
    inputs -- with the kernel that ran asserted (a dispatch change must not silently move these shapes elsewhere)."""
    from thinktwice_amd import ops, weights
    N, H, W, Cin, Cout, k, stride, act, use_bn, res, window, kern = case
    g = torch.Generator().manual_seed(Cin * 7 + Cout + H)
    pad = k // 2
    x = _mk((N, Cin, H, W), g)
    w = _mk((Cout, Cin, k, k), g, (Cin * k * k) ** -0.5)
    scale = (torch.rand(Cout, generator=g) + 0.5) if use_bn else None
    shift = _mk((Cout,), g, 0.3)
    xq = weights.to_channel_last(x, torch.float32)
    in_coff = 0
    if window:       # the layer reads channels [32, 32 + Cin) of a (N, H, W, Cin + 64) buffer (torch.cat that never copies)
        wide = _mk((N, H, W, Cin + 64), g)
        wide[..., 32:32 + Cin] = xq
        xq, in_coff = wide, 32
    xq = xq.cuda()
    wq = weights.prep_conv_weight(w, torch.float32).cuda()
    wx = weights.split_pairs_x3(wq)
    ref = F.conv2d(x, w, None, stride, pad)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    ref = ref + shift.view(1, -1, 1, 1)
    rs = []
    for _ in range(res):
        r = _mk(tuple(ref.shape), g).permute(0, 2, 3, 1).contiguous()
        ref = ref + r.permute(0, 3, 1, 2)
        rs.append(r.cuda())
    ref = {0: lambda t: t, 1: F.relu, 3: F.gelu}[act](ref)
    kw = dict(stride=stride, pad=pad, scale=None if scale is None else scale.cuda(), shift=shift.cuda(), act=act,
              res1=rs[0] if res >= 1 else None, res2=rs[1] if res >= 2 else None, in_coff=in_coff, cin=Cin)
    out = ops.conv2d(xq, wq, w_x3=wx, **kw)
    assert ops._last_conv_kernel().replace(" + tail", "") == kern, ops._last_conv_kernel()
    got = out.cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    assert err < 1e-4, err
    exact = ops.conv2d(xq, wq, **kw)
    e2 = float((out - exact).abs().max() / exact.abs().max())
    assert 0 < e2 < 5e-5, e2
    assert torch.equal(out, ops.conv2d(xq, wq, w_x3=wx, **kw))        # repeatable: no race in the pipeline


def test_pipelined_tiles_are_bit_identical_to_the_8wave_tiles():
    """tools/x3_pipe_ab.py: every arm in its own process (the knobs are read once), outputs compared bit for bit -- 256-wide
    (8-wave tile vs the pipelined / run-staged kernels) and 128-wide -- plus repeatability and the exact-f32 distance."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for env_extra, arms in (({}, "0,1"), ({"TT_AB_SET": "128"}, "0,1"), ({"TT_X3_RUN3": "0"}, "0,1")):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "x3_pipe_ab.py"), "1", arms], cwd=root,
                           env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "ALL OK" in r.stdout, (env_extra, r.stdout[-1500:], r.stderr[-500:])


SPLITK_X3_CASES = [   # (N, H, W, Cin, Cout, k, act, use_bn, residuals, window)
    (8, 14, 28, 512, 512, 3, 1, True, 1, False),     # ResNet layer 4 of a batch-1 tick: M = 3136, 104 tiles x 4 K ranges
    (8, 14, 28, 2048, 512, 1, 1, True, 1, False),    # its 1x1 with K = 2048
    (3, 13, 29, 256, 320, 3, 0, False, 0, False),    # ragged M = 1131 (4.4 row tiles), 5 column tiles, 9 ranges of 8 K tiles
    (8, 14, 28, 512, 96, 3, 3, True, 2, True),       # Cout = 96: half-empty second column tile; channel-window input, GELU, 2 residuals
    (16, 4, 8, 512, 512, 3, 1, True, 0, False),      # M = 512 (the small-golden trunk): 16 tiles x 16 ranges
]


@pytest.mark.parametrize("case", SPLITK_X3_CASES, ids=[f"M{c[0] * c[1] * c[2]}-K{c[3] * c[5] * c[5]}-N{c[4]}" for c in SPLITK_X3_CASES])
def test_bf16x3_splitk_tile_matches_torch_f32(case):
    """Few rows, long K, bf16x3 operand: the 64-wide bf16x3 tile with the K tiles dealt over workgroups (conv_igemm_glds.hip
    x3_splitk_plan) + the ordered finalize kernel, against torch f32 (1e-4 of the output's max) and the exact-f32 split-K kernel
    the same layers ran before; the kernel that ran is asserted, and the result is run-to-run bit-identical (ordered slices)."""
    from thinktwice_amd import ops, weights
    N, H, W, Cin, Cout, k, act, use_bn, res, window = case
    g = torch.Generator().manual_seed(Cin + Cout + H)
    pad = k // 2
    x = _mk((N, Cin, H, W), g)
    w = _mk((Cout, Cin, k, k), g, (Cin * k * k) ** -0.5)
    scale = (torch.rand(Cout, generator=g) + 0.5) if use_bn else None
    shift = _mk((Cout,), g, 0.3)
    xq = weights.to_channel_last(x, torch.float32)
    in_coff = 0
    if window:
        wide = _mk((N, H, W, Cin + 64), g)
        wide[..., 32:32 + Cin] = xq
        xq, in_coff = wide, 32
    xq = xq.cuda()
    wq = weights.prep_conv_weight(w, torch.float32).cuda()
    wx = weights.split_pairs_x3(wq)
    ref = F.conv2d(x, w, None, 1, pad)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    ref = ref + shift.view(1, -1, 1, 1)
    rs = []
    for _ in range(res):
        r = _mk(tuple(ref.shape), g).permute(0, 2, 3, 1).contiguous()
        ref = ref + r.permute(0, 3, 1, 2)
        rs.append(r.cuda())
    ref = {0: lambda t: t, 1: F.relu, 3: F.gelu}[act](ref)
    kw = dict(stride=1, pad=pad, scale=None if scale is None else scale.cuda(), shift=shift.cuda(), act=act,
              res1=rs[0] if res >= 1 else None, res2=rs[1] if res >= 2 else None, in_coff=in_coff, cin=Cin)
    out = ops.conv2d(xq, wq, w_x3=wx, **kw)
    assert ops._last_conv_kernel() == "conv_igemm_glds_kernel<float, 64, 8, 1, 128, 2, false, true> split-K", ops._last_conv_kernel()
    again = ops.conv2d(xq, wq, w_x3=wx, **kw)
    got = out.cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape and torch.equal(out, again)
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    assert err < 1e-4, err
    exact = ops.conv2d(xq, wq, **kw)
    assert "split-K" in ops._last_conv_kernel() and "conv_igemm_kernel" in ops._last_conv_kernel(), ops._last_conv_kernel()
    err = float((out - exact).abs().max() / exact.abs().max().clamp_min(1e-6))
    assert err < 1e-4, err


PAIR_CASES = [
    # N, H, W, Cin, Cout, k, stride, act   (M > 4096: the compiler-scheduled LDS-DMA bf16x3 tiles, 32 / 64 / 128 / 256 wide)
    (3, 40, 48, 64, 64, 3, 1, 1),        # layer1's 3 x 3 (64-wide tile), ragged M
    (2, 64, 64, 128, 64, 3, 1, 1),       # unet_layer0.1
    (3, 300, 300, 64, 12, 3, 1, 0),      # fused seg head (32-wide tile)
    (2, 50, 60, 64, 256, 1, 1, 0),       # 256-wide 8-wave tile
    (4, 56, 100, 96, 128, 3, 2, 1),      # stride 2, three channel chunks (96 = 3 x 32), M = 5600
    # K >= 1152: the hand-pipelined kernels (csrc/conv_x3_pipe.hip), pre-split variants
    (2, 64, 64, 128, 128, 3, 1, 1),      # run-staged 3 x 3, 128-wide
    (4, 40, 48, 256, 256, 3, 1, 1),      # run-staged 3 x 3, 256-wide, ragged M (7680)
    (4, 56, 100, 256, 256, 3, 2, 0),     # per-tap pipeline (stride 2), 256-wide
    (2, 50, 60, 2048, 512, 1, 1, 1),     # 1 x 1 with K = 2048 (ASPP conv1), per-tap pipeline
    (4, 56, 100, 128, 128, 3, 2, 1),     # per-tap pipeline, 128-wide
]


@pytest.mark.parametrize("case", PAIR_CASES)
def test_conv_pair_format_activations_are_bit_identical(case):
    """tt_conv_desc.in_pair / out_pair: moving the bf16 (hi, lo) operand split from the consumer's K loop to the producer changes
    nothing in the sums -- a convolution reading the pre-split tensor equals the one reading f32 bit for bit, and a convolution
    writing pair format writes exactly the split of what it would have written."""
    from thinktwice_amd import ops, weights
    N, H, W, Cin, Cout, k, stride, act = case
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = _mk((N, Cin, H, W), g)
    w = _mk((Cout, Cin, k, k), g, (Cin * k * k) ** -0.5)
    shift = _mk((Cout,), g, 0.3).cuda()
    xq = weights.to_channel_last(x, torch.float32).cuda()
    wq = weights.prep_conv_weight(w, torch.float32).cuda()
    wx = weights.split_pairs_x3(wq)
    xp = weights.split_pairs_x3(xq)                       # per 16 channels: [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15]
    ref = ops.conv2d(xq, wq, stride=stride, pad=k // 2, shift=shift, act=act, w_x3=wx)
    got = ops.conv2d(xp, wq, stride=stride, pad=k // 2, shift=shift, act=act, w_x3=wx, in_pair=True)
    assert "pre-split A" in ops._last_conv_kernel()
    assert torch.equal(got, ref)
    if Cout % 16 == 0:
        outp = ops.conv2d(xq, wq, stride=stride, pad=k // 2, shift=shift, act=act, w_x3=wx, out_pair=True)
        assert torch.equal(outp.view(torch.int32), weights.split_pairs_x3(ref.contiguous()).view(torch.int32))
        both = ops.conv2d(xp, wq, stride=stride, pad=k // 2, shift=shift, act=act, w_x3=wx, in_pair=True, out_pair=True)
        assert torch.equal(both.view(torch.int32), outp.view(torch.int32))


def test_bilinear_up2_pair_format_is_the_split_of_the_f32_result():
    from thinktwice_amd import ops, weights
    g = torch.Generator().manual_seed(9)
    x = _mk((3, 17, 23, 128), g).cuda()
    ref = ops.bilinear_up2(x)
    got = ops.bilinear_up2(x, out_pair=True)
    assert torch.equal(got.view(torch.int32), weights.split_pairs_x3(ref).view(torch.int32))


def test_pair_format_is_refused_outside_the_kernel_contract():
    from thinktwice_amd import _lib, ops, weights
    x = torch.zeros(1, 8, 8, 64).cuda()                      # 64 rows: below the LDS-DMA kernel's row count
    w = torch.zeros(64, 1, 1, 64).cuda()
    with pytest.raises(_lib.TTError):
        ops.conv2d(x, w, w_x3=weights.split_pairs_x3(w), in_pair=True)


H2_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, act, bn, res, out_f32   ("h2": half storage x f16 (hi, lo) weights, csrc/conv_h2.hip)
    (3, 40, 48, 64, 128, 3, 1, 1, 1, 1, True, 1, False),    # 128-wide tile, ragged M (5760), half residual + ReLU, 9 K tiles
    (2, 64, 64, 128, 64, 1, 1, 0, 1, 0, False, 0, False),   # 64-wide tile, 1x1, two K tiles
    (2, 56, 100, 256, 512, 3, 2, 1, 1, 1, True, 0, True),   # stride 2, four column tiles, f32 output
    (2, 50, 60, 64, 256, 1, 1, 0, 1, 1, True, 1, False),    # ONE K tile (layer1's 1x1 convs), residual
    (5, 30, 30, 192, 200, 3, 1, 1, 1, 0, False, 0, True),   # Cout tail (200) in the 128 tile
    (2, 21, 21, 512, 256, 3, 1, 1, 1, 1, True, 0, False),   # few rows (M = 882): the same kernel, partial row tile
    (1, 9, 7, 2048, 256, 1, 1, 0, 1, 0, False, 0, False),   # M = 63 (PAFPN lateral 3 of a small image)
    (2, 40, 48, 64, 64, 3, 1, 6, 6, 1, True, 2, False),     # dilation, two residuals (rolled epilogue)
]


@pytest.mark.parametrize("case", H2_CASES)
def test_conv_h2_matches_torch_f32(case):
    """Layer mode "h2" (the PAFPN of the mixed mode "f32x3h"): IEEE-half activations as stored times an f16 (hi, lo) weight pair,
    two f16 MFMAs per product.  With respect to the STORED activations the product is exact to ~2^-22: within 2e-5 (of the
    output's max) of the f32 reference evaluated on the half-rounded input and the unrounded weights; a half output adds its own
    rounding (2^-11 relative per element)."""
    from thinktwice_amd import ops, weights
    N, H, W, Cin, Cout, k, stride, pad, dil, act, use_bn, res, out_f32 = case
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = _mk((N, Cin, H, W), g)
    w = _mk((Cout, Cin, k, k), g, (Cin * k * k) ** -0.5)
    scale = (torch.rand(Cout, generator=g) + 0.5) if use_bn else None
    shift = _mk((Cout,), g, 0.3)
    xq = weights.to_channel_last(x, torch.float16).cuda()
    w32 = weights.prep_conv_weight(w, torch.float32).cuda()
    wh = weights.split_pairs_h2(w32)
    assert wh.dtype == torch.float16 and wh.shape[-1] == 2 * w32.shape[-1]
    ref = F.conv2d(x.half().float(), w, None, stride, pad, dil)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    ref = ref + shift.view(1, -1, 1, 1)
    r1 = r2 = None
    if res >= 1:
        r1 = _mk(tuple(ref.shape), g).permute(0, 2, 3, 1).contiguous().half()
        ref = ref + r1.float().permute(0, 3, 1, 2)
    if res >= 2:
        r2 = _mk(tuple(ref.shape), g).permute(0, 2, 3, 1).contiguous().half()
        ref = ref + r2.float().permute(0, 3, 1, 2)
    ref = {0: lambda t: t, 1: F.relu}[act](ref)
    out2 = torch.zeros(N, ref.shape[2], ref.shape[3], Cout + 8, device="cuda") if (not out_f32 and res < 2) else None
    out = ops.conv2d(xq, w32.half(), stride=stride, pad=pad, dil=dil, scale=None if scale is None else scale.cuda(),
                     shift=shift.cuda(), act=act, res1=None if r1 is None else r1.cuda(),
                     res2=None if r2 is None else r2.cuda(), w_h2=wh, out_dtype=torch.float32 if out_f32 else None,
                     out2=out2, out2_coff=8)
    assert out.dtype == (torch.float32 if out_f32 else torch.float16)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    scale_ = ref.abs().max().clamp_min(1e-6)
    if out_f32:
        assert float((got - ref).abs().max() / scale_) < 2e-5
    else:
        assert torch.equal(got, ref.half().float()) or float((got - ref).abs().max() / scale_) < 6e-4
        # the half output is the f32 result rounded once: all but a few ties-at-the-last-bit agree with rounding the reference
        assert float((got != ref.half().float()).float().mean()) < 0.02
    if out2 is not None:     # the second, f32 copy carries the unrounded values
        got2 = out2[..., 8:].cpu().permute(0, 3, 1, 2)
        assert float((got2 - ref).abs().max() / scale_) < 2e-5
        assert float(out2[..., :8].abs().max()) == 0.0


def test_h2_pipelined_kernel_is_bit_identical_to_the_8wave_kernel():
    """tools/h2_pipe_ab.py: conv_h2_pipe_kernel (long-K 128-wide layers: the PAFPN's 3 x 3 convolutions) against conv_h2_kernel
    (TT_H2_PIPE=0), each arm in its own process, outputs compared bit for bit; plus repeatability and the distance to torch f32."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "h2_pipe_ab.py"), "1"], cwd=root, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, (r.stdout[-2500:], r.stderr[-500:])


def test_bf16x3_layer_can_store_half():
    """A bf16x3 layer (f32 operands) feeding a half-storage stage writes IEEE half directly (stem -> layer1, PAFPN laterals 2 / 3)."""
    from thinktwice_amd import ops, weights
    g = torch.Generator().manual_seed(5)
    x = _mk((2, 64, 40, 48), g)
    w = _mk((256, 64, 1, 1), g, 64 ** -0.5)
    b = _mk((256,), g, 0.2)
    xq = weights.to_channel_last(x, torch.float32).cuda()
    wq = weights.prep_conv_weight(w, torch.float32).cuda()
    ref = F.conv2d(x, w, b)
    for wx in (weights.split_pairs_x3(wq), None):
        out = ops.conv2d(xq, wq, shift=b.cuda(), w_x3=wx, out_dtype=torch.float16)
        assert out.dtype == torch.float16
        got = out.float().cpu().permute(0, 3, 1, 2)
        assert float((got - ref).abs().max() / ref.abs().max()) < 6e-4
        assert float((got != ref.half().float()).float().mean()) < 0.02


@pytest.mark.parametrize("shape", [(2, 56, 100, 28, 50), (3, 45, 37, 23, 19)], ids=["x2", "odd"])
@pytest.mark.parametrize("mode", ["f32", "x3"])
def test_conv_with_an_upsampled_residual_equals_conv_then_upsample_add(shape, mode):
    """tt_conv_desc.res1_up_h / _w: the PAFPN top-down sum `lat[i-1] += F.interpolate(lat[i], size=..., mode='nearest')`
    (backbones/lss.py:301-305) inside the lateral conv's epilogue is bit-identical to the conv followed by tt_upsample_nearest_add,
    and both equal torch's interpolate."""
    from thinktwice_amd import ops, weights
    N, H, W, h, w = shape
    g = torch.Generator().manual_seed(H + W)
    x = _mk((N, H, W, 64), g).cuda()
    wt = _mk((256, 1, 1, 64), g, 64 ** -0.5).cuda()
    b = _mk((256,), g, 0.2).cuda()
    coarse = _mk((N, h, w, 256), g).cuda()
    wx = weights.split_pairs_x3(wt) if mode == "x3" else None
    fused = ops.conv2d(x, wt, shift=b, w_x3=wx, res1=coarse, res1_up=True)
    plain = ops.conv2d(x, wt, shift=b, w_x3=wx)
    want = plain.clone()
    ops.upsample_nearest_add_(want, coarse)
    assert torch.equal(fused, want)
    up = F.interpolate(coarse.permute(0, 3, 1, 2), size=(H, W), mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(want, plain + up)


@pytest.mark.parametrize("out_half", [True, False], ids=["half-out", "f32-out"])
def test_conv_h2_with_an_f32_residual(out_half):
    """tt_conv_desc.res1_f32: the mixed mode's PAFPN keeps its bottom-up SUM in f32 and feeds the 3 x 3 convolutions half copies --
    the h2 layer adds an f32 residual and writes half (the next conv's input) or f32 (the sum itself)."""
    from thinktwice_amd import ops, weights
    g = torch.Generator().manual_seed(21)
    x = _mk((3, 64, 40, 48), g)
    w = _mk((256, 64, 3, 3), g, (64 * 9) ** -0.5)
    b = _mk((256,), g, 0.3)
    res = _mk((3, 20, 24, 256), g).cuda()                                        # f32, stride-2 output geometry
    xq = weights.to_channel_last(x, torch.float16).cuda()
    w32 = weights.prep_conv_weight(w, torch.float32).cuda()
    ref = F.conv2d(x.half().float(), w, b, 2, 1) + res.cpu().permute(0, 3, 1, 2)
    out = ops.conv2d(xq, w32.half(), stride=2, pad=1, shift=b.cuda(), res1=res, w_h2=weights.split_pairs_h2(w32),
                     out_dtype=torch.float16 if out_half else torch.float32)
    got = out.float().cpu().permute(0, 3, 1, 2)
    scale = float(ref.abs().max())
    if out_half:
        assert out.dtype == torch.float16 and float((got != ref.half().float()).float().mean()) < 0.02
        assert float((got - ref).abs().max()) < 6e-4 * scale
    else:
        assert float((got - ref).abs().max()) < 2e-5 * scale


def test_bf16x3_layer_with_residual_can_store_half_and_an_f32_twin():
    """PAFPN lateral of the mixed mode: a bf16x3 1 x 1 conv (f32 in) adds the upsampled f32 top-down residual and writes BOTH the
    half copy the next 3 x 3 reads (primary output) and the f32 sum the next level's residual reads (tt_conv_desc.out2)."""
    from thinktwice_amd import ops, weights
    g = torch.Generator().manual_seed(22)
    x = _mk((2, 56, 100, 64), g).cuda()
    wt = _mk((256, 1, 1, 64), g, 64 ** -0.5).cuda()
    b = _mk((256,), g, 0.2).cuda()
    coarse = _mk((2, 28, 50, 256), g).cuda()
    wx = weights.split_pairs_x3(wt)
    want = ops.conv2d(x, wt, shift=b, w_x3=wx, res1=coarse, res1_up=True)        # the f32 form (tested above)
    twin = torch.empty_like(want)
    half = ops.conv2d(x, wt, shift=b, w_x3=wx, res1=coarse, res1_up=True, out_dtype=torch.float16, out2=twin)
    assert half.dtype == torch.float16
    assert torch.equal(twin, want)
    assert torch.equal(half, want.half())
