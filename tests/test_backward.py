"""Training-step backward through the tape (thinktwice_amd/autodiff.py): the ResNet-50 + PAFPN camera trunk's
parameter gradients against torch autograd through the oracle (oracle/model_ref.py resnet50 / pafpn, which reproduce the
reference's modules) under a synthetic loss  L = sum_k <fpn_k, R_k>."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("x3,train", [(False, False), (True, False), (False, True)])
def test_camera_trunk_backward_matches_oracle_autograd(x3, train):
    """`train`: model.train() semantics -- batch-statistics BatchNorm in the ResNet (oracle inside train_ref.train_mode())."""
    from oracle import model_ref as M, train_ref as TR
    from thinktwice_amd import autodiff, config, layers, params, weights
    import contextlib
    hw, NI = (64, 128), 4 if train else 2
    mode = TR.train_mode if train else contextlib.nullcontext
    cfg = config.model_config(final_dim=hw)
    sd = params.init_params(cfg, seed=3, parts=("img_encoder",))
    # ---- oracle: autograd over the trunk's parameters
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if (k.startswith("img_encoder.img_backbone") or k.startswith("img_encoder.img_neck"))
              and v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    sdr = dict(sd)
    sdr.update(leaves)
    g = torch.Generator().manual_seed(7)
    img = torch.randn(NI, 3, *hw, generator=g)
    with mode():
        outs = M.pafpn(sdr, "img_encoder.img_neck", M.resnet50(sdr, "img_encoder.img_backbone", img))
        R = [torch.randn(o.shape, generator=g) for o in outs]
        loss = sum((o * r).sum() for o, r in zip(outs, R))
        loss.backward()
    # ---- HIP: taped forward of the same trunk, seeded with R, backward kernels
    from thinktwice_amd.lss import LSS
    enc_cfg = {k: v for k, v in cfg["img_encoder"].items() if k != "type"}
    enc = LSS(**enc_cfg, dtype="f32x3" if x3 else torch.float32).load_state_dict(sd)
    x = weights.to_channel_last(img, torch.float32).cuda()
    saved, layers.BN_TRAIN = layers.BN_TRAIN, train
    try:
        with autodiff.Tape(x3=x3) as tape:
            bufs = enc._trunk(x)
            for (t, off, c), r in zip(enc._fpn_views(bufs), R):
                tape.seed(t[..., off:off + c], r.permute(0, 2, 3, 1))
            tape.backward()
    finally:
        layers.BN_TRAIN = saved
    torch.cuda.synchronize()
    # forward sanity (same maps), then every parameter's gradient
    for (t, off, c), o in zip(enc._fpn_views(bufs), outs):
        got = t[..., off:off + c].permute(0, 3, 1, 2).cpu()
        assert float((got - o.detach()).abs().max() / o.detach().abs().max()) < (2e-4 if x3 else 2e-5)
    missing = [k for k in leaves if leaves[k].grad is not None and k not in tape.param_grads]
    assert not missing, missing[:5]
    tol = 3e-3 if x3 else (1e-3 if train else 2e-4)
    worst = {}
    for k, v in leaves.items():
        if v.grad is None:
            continue
        if train and k.endswith(".bias") and ".conv" in k:
            continue            # (a conv bias in front of a batch-statistics BatchNorm: zero gradient + rounding noise)
        got = tape.param_grads[k].cpu()
        assert got.shape == v.grad.shape, (k, got.shape, v.grad.shape)
        worst[k] = float((got - v.grad).abs().max() / v.grad.abs().max().clamp_min(1e-12))
    bad = {k: e for k, e in worst.items() if e > tol}
    print("trunk backward: params", len(worst), "worst rel err", max(worst.values()), "top",
          [(k[25:], round(e, 5), float(leaves[k].grad.abs().max())) for k, e in sorted(worst.items(), key=lambda kv: -kv[1])[:12]])
    if train:
        # one ReLU mask flips on this input (channel 10 of layer1.0.conv1: tools/debug_trunk_train.py shows the whole excess in
        # that one output channel; the same bottleneck is exact to 5e-7 in isolation, tools/debug_bn_block.py): its own
        # weight / bias gradients move by ~1e-2, everything else stays inside 1e-3
        assert len(bad) <= 3 and max(bad.values(), default=0.0) < 3e-2, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
        assert float(np.median(list(worst.values()))) < 2e-5
    else:
        assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]


def test_seg_loss_backward_through_unet_and_trunk_matches_oracle_autograd():
    """A whole loss term end to end: focal segmentation loss (encoder_decoder_framework.py:172-176) -> UNet (transposed
    convs, concat windows, x2 bilinear) -> PAFPN -> ResNet-50: d(seg_loss)/d(parameter) for every parameter on that path
    against loss.backward() through the oracle."""
    from oracle import model_ref as M, train_ref as TR
    from thinktwice_amd import autodiff, config, params, weights
    from thinktwice_amd.losses import LossReducer
    from thinktwice_amd.lss import LSS
    hw, B, N = (64, 128), 1, 2
    cfg = config.model_config(final_dim=hw)
    sd = params.init_params(cfg, seed=5, parts=("img_encoder",))
    pre = ("img_encoder.img_backbone", "img_encoder.img_neck", "img_encoder.seg_net")
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if k.startswith(pre) and v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    sdr = dict(sd)
    sdr.update(leaves)
    g = torch.Generator().manual_seed(9)
    img = torch.randn(B * N, 3, *hw, generator=g)
    labels = torch.randint(0, 12, (B, N, *hw), generator=g).float()
    labels[torch.rand(B, N, *hw, generator=g) < 0.05] = 255.0
    feats = M.pafpn(sdr, "img_encoder.img_neck", M.resnet50(sdr, "img_encoder.img_backbone", img))
    loss = TR.seg_loss(M.unet(sdr, "img_encoder.seg_net", feats), labels)
    loss.backward()

    enc_cfg = {k: v for k, v in cfg["img_encoder"].items() if k != "type"}
    enc = LSS(**enc_cfg, dtype="f32x3").load_state_dict(sd)
    red = LossReducer("cuda")
    x = weights.to_channel_last(img, torch.float32).cuda()
    with autodiff.Tape(x3=True) as tape:
        seg = enc._seg_net(enc._trunk(x))
        got_loss = red.seg_focal(seg, labels, num_classes=12, factor=2)
        tape.seed(seg, red.seg_focal_bwd(seg, labels, num_classes=12, factor=2))
        tape.backward()
    torch.cuda.synchronize()
    assert abs(float(got_loss) - float(loss.detach())) < 1e-4 * abs(float(loss.detach()))
    worst = {}
    for k, v in leaves.items():
        if v.grad is None:
            continue
        assert k in tape.param_grads, k
        got = tape.param_grads[k].cpu()
        assert got.shape == v.grad.shape, (k, got.shape, v.grad.shape)
        worst[k] = float((got - v.grad).abs().max() / v.grad.abs().max().clamp_min(1e-12))
    print("seg-loss backward: params", len(worst), "worst rel err", max(worst.values()))
    bad = {k: e for k, e in worst.items() if e > 3e-3}
    assert len(worst) > 200 and not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]


def test_depth_loss_backward_through_depthnet_and_trunk_matches_oracle_autograd():
    """Depth BCE (encoder_decoder_framework.py:179-190) + a synthetic term on the context features -> DepthNet (camera-aware
    SE gates from the BatchNorm1d'd camera vector, BasicBlocks, ASPP with its image-pooling branch folded into a per-image
    shift, deformable conv incl. its offset branch) -> neck_conv -> PAFPN -> ResNet-50, against loss.backward() through
    the oracle."""
    from oracle import model_ref as M, train_ref as TR
    from thinktwice_amd import autodiff, config, params, weights
    from thinktwice_amd.losses import LossReducer
    from thinktwice_amd.lss import LSS
    hw, B, N = (64, 128), 1, 2
    cfg = config.model_config(final_dim=hw)
    sd = params.init_params(cfg, seed=6, parts=("img_encoder",))
    p = "img_encoder"
    pre = (p + ".img_backbone", p + ".img_neck", p + ".neck_conv", p + ".depth_net")
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if k.startswith(pre) and v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
    sdr = dict(sd)
    sdr.update(leaves)
    g = torch.Generator().manual_seed(10)
    img = torch.randn(B * N, 3, *hw, generator=g)
    mlp_in = torch.randn(B * N, 22, generator=g)
    gt_depth = torch.rand(B, N, *hw, generator=g) * 45.0
    gt_depth[torch.rand(B, N, *hw, generator=g) < 0.9] = 0.0
    d_bound = cfg["img_encoder"]["d_bound"]
    D = int((d_bound[1] - d_bound[0]) / d_bound[2])
    fpn = M.pafpn(sdr, p + ".img_neck", M.resnet50(sdr, p + ".img_backbone", img))
    df = M.depth_net(sdr, p + ".depth_net", M.conv(sdr, p + ".neck_conv", fpn[2]), mlp_in)
    Rc = torch.randn(df[:, D:D + 256].shape, generator=g) * 0.01
    loss = TR.depth_loss(df[:, :D], gt_depth, d_bound, 16) + (df[:, D:D + 256] * Rc).sum()
    loss.backward()

    enc_cfg = {k: v for k, v in cfg["img_encoder"].items() if k != "type"}
    enc = LSS(**enc_cfg, dtype="f32x3").load_state_dict(sd)
    red = LossReducer("cuda")
    x = weights.to_channel_last(img, torch.float32).cuda()
    with autodiff.Tape(x3=True) as tape:
        bufs = enc._trunk(x)
        fb, foff, _ = enc._fpn_views(bufs)[2]
        src = enc.neck_conv(fb, in_coff=foff, cin=256)
        depth, merge_in = enc._depth_net(src, mlp_in.cuda(), 1)
        got = red.depth_bce(depth, gt_depth, d_bound, 16)
        tape.seed(depth, red.depth_bce_bwd(depth, gt_depth, d_bound, 16))
        tape.seed(merge_in[..., :256], Rc.permute(0, 2, 3, 1))
        tape.backward()
    torch.cuda.synchronize()
    want_depth = TR.depth_loss(df[:, :D], gt_depth, d_bound, 16)
    assert abs(float(got) - float(want_depth.detach())) < 1e-4 * abs(float(want_depth.detach()))
    worst = {}
    for k, v in leaves.items():
        if v.grad is None:
            continue
        assert k in tape.param_grads, k
        got_g = tape.param_grads[k].cpu()
        assert got_g.shape == v.grad.shape, (k, got_g.shape, v.grad.shape)
        worst[k] = float((got_g - v.grad).abs().max() / v.grad.abs().max().clamp_min(1e-12))
    print("depth-loss backward: params", len(worst), "worst rel err", max(worst.values()))
    bad = {k: e for k, e in worst.items() if e > 3e-3}
    assert len(worst) > 240 and not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


# Metric: relative L2 error per parameter (bound `tol`), plus a loose bound on the worst single element.  Most parameters
# agree to ~1e-6 element-wise; a few (layer3 / layer4 BatchNorm and conv parameters, and with 4.5e-4 relative L2 everything
# upstream of them down to the stem) do not, identically in the exact-f32 and the bf16x3 mode: the product path applies eval
# BatchNorm as a folded affine (conv * scale + shift) where torch normalises first, the two round differently at the 1e-7
# level, and the handful of pre-activations that close to 0 get a different ReLU mask.  On the 2x4 .. 4x8-pixel maps of this
# test one flipped pixel is percents of one channel's sum (checked: every kernel involved is exact to 1e-6 on these shapes,
# tools/small_map_conv_check.py, and the same trunk with 2 images instead of 8 agrees to 3e-6 throughout).
# (the op-level and sub-network tests above hold 1e-6..4e-6; this test checks the wiring of the whole encoder.  With the
# atomic split-K off the forward is run-to-run identical and so are these numbers: exact f32 4.5e-4 relative L2 (worst element
# 2.2e-3); bf16x3 1.1e-2 on the seg-to-feature parameters, whose ReLUs see the 12 segmentation logits with the forward's
# ~1e-5 error -- a hundred times more masks within rounding of 0 than in f32 -- on 16 x 32-pixel and smaller maps.  Bounds:
# about 3x those.)
@pytest.mark.parametrize("mode,tol", [("f32", 1.5e-3), ("f32x3", 3e-2)])
def test_whole_camera_encoder_backward_matches_oracle_autograd(mode, tol, monkeypatch):
    """LSS.forward under the tape, two sweeps x four cameras: BEV (synthetic upstream gradient), focal segmentation loss and
    depth BCE together; every parameter of `img_encoder` against loss.backward() through oracle.lss_forward -- including the
    reference's gradient stops (older sweeps under no_grad lss.py:711, seg logits detached into seg-to-feature lss.py:589)."""
    from oracle import model_ref as M, train_ref as TR
    from thinktwice_amd import autodiff, config, params, synth
    from thinktwice_amd import ops
    from thinktwice_amd.losses import LossReducer
    from thinktwice_amd.lss import LSS
    hw, B = (64, 128), 1
    cfg = config.model_config(final_dim=hw)
    sd = params.init_params(cfg, seed=8, parts=("img_encoder",))
    batch = synth.make_batch(B, img_hw=hw, num_points=1000)
    tgt = synth.make_train_targets(B, img_hw=hw)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if k.startswith("img_encoder.") and v.is_floating_point() and v.dim() > 0
              and not k.endswith(("running_mean", "running_var", "frustum", "voxel_size", "voxel_coord", "voxel_num"))}
    sdr = dict(sd)
    sdr.update(leaves)
    ref = M.lss_forward(sdr, "img_encoder", cfg, batch["img"], batch["img_metas"])
    g = torch.Generator().manual_seed(12)
    Rb = torch.randn(ref["bev"].shape, generator=g) * 0.05
    d_bound = cfg["img_encoder"]["d_bound"]
    loss = (ref["bev"] * Rb).sum() + TR.seg_loss(ref["seg"], tgt["seg"]) + TR.depth_loss(ref["depth"], tgt["depth"], d_bound, 16)
    loss.backward()

    enc_cfg = {k: v for k, v in cfg["img_encoder"].items() if k != "type"}
    enc = LSS(**enc_cfg, dtype="f32x3" if mode == "f32x3" else torch.float32).load_state_dict(sd)
    red = LossReducer("cuda")
    with autodiff.Tape(x3=(mode == "f32x3")) as tape:
        out = enc(batch["img"].cuda(), batch["img_metas"], channel_last=True)
        seg, depth, bev = out["_seg_cl"], out["_depth_cl"], out["_bev_cl"]
        red.seg_focal(seg, tgt["seg"], num_classes=12, factor=2)
        tape.seed(seg, red.seg_focal_bwd(seg, tgt["seg"], num_classes=12, factor=2))
        red.depth_bce(depth, tgt["depth"], d_bound, 16)
        tape.seed(depth, red.depth_bce_bwd(depth, tgt["depth"], d_bound, 16))
        tape.seed(bev, Rb.permute(0, 2, 3, 1))
        tape.backward()
    torch.cuda.synchronize()
    worst, missing = {}, []
    for k, v in leaves.items():
        if v.grad is None or float(v.grad.abs().max()) == 0.0:
            continue
        if k not in tape.param_grads:
            missing.append(k)
            continue
        got = tape.param_grads[k].cpu()
        assert got.shape == v.grad.shape, (k, got.shape, v.grad.shape)
        worst[k] = (float((got - v.grad).norm() / v.grad.norm().clamp_min(1e-20)),
                    float((got - v.grad).abs().max() / v.grad.abs().max().clamp_min(1e-12)))
    assert not missing, missing[:10]
    print("camera encoder backward: params", len(worst), "worst L2 rel", max(e[0] for e in worst.values()),
          "worst element rel", max(e[1] for e in worst.values()))
    bad = {k: e for k, e in worst.items() if e[0] > tol or e[1] > 6 * tol}
    assert len(worst) > 280 and not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:10]


@pytest.mark.parametrize("train", [False, True])
def test_lidar_encoder_backward_matches_oracle_autograd(train):
    """`train`: model.train() semantics (batch-statistics BatchNorm1d over the live rows of every sparse level, BatchNorm2d in
    SECOND / SECONDFPN).  LidarNet under the tape: sparse encoder (submanifold + strided rulebook convs with BatchNorm1d, residual blocks),
    dense conversion, SECOND blocks, SECONDFPN (1x1 conv + transposed conv with BN/ReLU); every parameter gradient against
    loss.backward() through oracle.lidar_net under  L = <out, R>."""
    import contextlib
    from oracle import model_ref as M, train_ref as TR
    from thinktwice_amd import autodiff, config, layers, params
    from thinktwice_amd.lidarnet import LidarNet
    import test_lidar
    mode = TR.train_mode if train else contextlib.nullcontext
    cfg = config.model_config()
    sd = params.init_params(cfg, seed=2, parts=("lidar_encoder",))
    pts = test_lidar._pts(2 if train else 1, 3000, seed=4)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if k.startswith("lidar_encoder.") and v.is_floating_point() and v.dim() > 0
              and not k.endswith(("running_mean", "running_var"))}
    sdr = dict(sd)
    sdr.update(leaves)
    with mode():
        ref = M.lidar_net(sdr, "lidar_encoder", cfg, pts)[0]
        g = torch.Generator().manual_seed(13)
        R = torch.randn(ref.shape, generator=g)
        (ref * R).sum().backward()

    le = dict(cfg["lidar_encoder"])
    le.pop("type")
    net = LidarNet(**le).load_state_dict(sd)
    saved, layers.BN_TRAIN = layers.BN_TRAIN, train
    try:
        with autodiff.Tape(x3=False) as tape:
            out = net(pts.cuda(), channel_last=True)
            tape.seed(out, R.permute(0, 2, 3, 1))
            tape.backward()
    finally:
        layers.BN_TRAIN = saved
    torch.cuda.synchronize()
    got_fwd = out.permute(0, 3, 1, 2).cpu()
    assert float((got_fwd - ref.detach()).abs().max() / ref.detach().abs().max()) < 1e-4
    worst, missing = {}, []
    for k, v in leaves.items():
        if v.grad is None or float(v.grad.abs().max()) == 0.0:
            continue
        if train and k.endswith(".bias") and k.rsplit(".", 1)[0] + ".weight" in leaves and leaves[k.rsplit(".", 1)[0] + ".weight"].dim() > 1:
            continue            # (a conv bias in front of a batch-statistics BatchNorm: zero gradient + rounding noise)
        if k not in tape.param_grads:
            missing.append(k)
            continue
        got = tape.param_grads[k].cpu()
        assert got.shape == v.grad.shape, (k, got.shape, v.grad.shape)
        worst[k] = (float((got - v.grad).norm() / v.grad.norm().clamp_min(1e-20)),
                    float((got - v.grad).abs().max() / v.grad.abs().max().clamp_min(1e-12)))
    assert not missing, missing[:10]
    print("lidar encoder backward: params", len(worst), "worst L2 rel", max(e[0] for e in worst.values()),
          "worst element rel", max(e[1] for e in worst.values()))
    # the forward itself agrees with the oracle to ~1e-4 of the map's max here (27-tap sums in a different order, BN1d
    # folded), so ReLU masks differ on more elements than in the camera trunk: 5e-4 .. 1.1e-3 relative L2 observed
    # train mode: batch statistics are computed in a different order (f64 partial sums here), so the forward agrees to ~1e-6
    # instead of ~1e-7 and a few more ReLU masks flip on these 3000-point clouds: 4.8e-3 observed on the BatchNorm biases
    lim = (8e-3, 4e-2) if train else (3e-3, 2e-2)
    bad = {k: e for k, e in worst.items() if e[0] > lim[0] or e[1] > lim[1]}
    assert len(worst) > 100 and not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:10]


@pytest.mark.parametrize("train", [False, True])
def test_fusion_neck_and_flatten_backward_matches_oracle_autograd(train):
    """`train`: model.train() semantics (batch-statistics BatchNorm2d / the BatchNorm1d of output_fc).  BEV fusion neck + SEBasicBlocks (mean/amax pooling, gated residual) + flatten tail (EDF:213-235, utils.py:84-121) under
    the tape, with upstream gradients on everything the losses touch: flat, the 32x21x21 map and the three coarser maps;
    parameter gradients AND the gradients handed back to the camera / LiDAR BEV inputs against oracle autograd."""
    import contextlib
    from oracle import model_ref as M, train_ref as TR
    from thinktwice_amd import autodiff, config, layers, params
    from thinktwice_amd.fusion import BEVFusion
    mode = TR.train_mode if train else contextlib.nullcontext
    B = 4 if train else 2
    cfg = config.model_config()
    sd = params.init_params(cfg, seed=4, parts=("fusion",))
    names = [k for k in sd if k.split(".")[0] in ("conv_cam", "conv_lidar", "conv_fusion", "_256_to_32", "MLP21", "MLP10",
                                                     "MLP4", "MLP2", "conv21_10", "conv10_4", "conv4_2", "output_fc")]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names
              if sd[k].is_floating_point() and sd[k].dim() > 0 and not k.endswith(("running_mean", "running_var"))}
    sdr = dict(sd)
    sdr.update(leaves)
    g = torch.Generator().manual_seed(14)
    cam = torch.randn(B, 256, 21, 21, generator=g).requires_grad_(True)
    lid = torch.randn(B, 512, 84, 84, generator=g).abs().requires_grad_(True)
    with mode():
        flat, f21, mids = M.fusion(sdr, cam, lid)
        outs = [flat, f21] + mids[3:]
        R = [torch.randn(o.shape, generator=g) for o in outs]
        sum((o * r).sum() for o, r in zip(outs, R)).backward()

    fus = BEVFusion(sd, "cuda")
    cl = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()   # noqa: E731
    camq, lidq = cl(cam), cl(lid)
    saved, layers.BN_TRAIN = layers.BN_TRAIN, train
    try:
        with autodiff.Tape(x3=False) as tape:
            hflat, hf21, hmids = fus(camq, lidq)
            tape.seed(hflat, R[0])
            for t, r in zip([hf21] + hmids[3:], R[1:]):
                tape.seed(t, r.permute(0, 2, 3, 1))
            tape.backward()
    finally:
        layers.BN_TRAIN = saved
    torch.cuda.synchronize()
    assert float((hflat.cpu() - flat.detach()).abs().max() / flat.detach().abs().max()) < 1e-4
    worst = {}
    for k, v in leaves.items():
        assert k in tape.param_grads, k
        got = tape.param_grads[k].cpu()
        assert got.shape == v.grad.shape, (k, got.shape, v.grad.shape)
        if train and k.endswith(".bias") and float(v.grad.norm()) < 1e-3 * float(leaves[k[:-4] + "weight"].grad.norm()):
            continue            # (a conv bias in front of a batch-statistics BatchNorm: zero gradient + rounding noise)
        worst[k] = float((got - v.grad).norm() / v.grad.norm().clamp_min(1e-20))
    for name, t, ref in (("d cam_bev", camq, cam), ("d lidar_bev", lidq, lid)):
        got = tape.grad(t).permute(0, 3, 1, 2).cpu()
        worst[name] = float((got - ref.grad).norm() / ref.grad.norm())
    print("fusion backward: tensors", len(worst), "worst L2 rel", max(worst.values()))
    # train mode: every block of this neck is exact against autograd on its own (tools/debug_se_train.py: 5e-7 per SE block,
    # 1e-4 through the 4-row BatchNorm1d of the whole tail); composed, a ReLU mask that flips on one of the 10x10 / 4x4 maps
    # (forward agreement 1e-6, pre-activations centred on 0 by the batch statistics) moves that block's gradients by ~1e-2
    lim = 2e-2 if train else 1e-3
    bad = {k: e for k, e in worst.items() if e > lim}
    assert len(worst) > 50 and not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
    if train:
        assert float(np.median(list(worst.values()))) < 3e-3


def test_spatial_gru_backward_matches_oracle_autograd():
    """SpatialGRU of the prediction module (dense_heads/utils.py:53-106): 4 steps x (update, reset, candidate) conv pairs with
    sigmoid gates, the gated blends and the decoder convs; parameter gradients and the gradients w.r.t. the BEV state and
    the (waypoint, control) inputs against oracle autograd."""
    from oracle import model_ref as M
    from thinktwice_amd import autodiff, config, params
    from thinktwice_amd.decoder import _GRU
    B, H, W = 2, 21, 21
    cfg = config.model_config()
    sd = params.init_params(cfg, seed=5, parts=("decoder",))
    p = "decoder.decoder_layers.0.prediction_module.spatial_gru"
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(p + ".")}
    sdr = dict(sd)
    sdr.update(leaves)
    g = torch.Generator().manual_seed(15)
    inp6 = torch.randn(B, 4, 6, generator=g).requires_grad_(True)
    state = torch.randn(B, 32, H, W, generator=g).requires_grad_(True)
    fut = M.spatial_gru(sdr, p, inp6[..., None, None].expand(B, 4, 6, H, W), state)       # (B, 4, 32, H, W)
    R = torch.randn(fut.shape, generator=g)
    (fut * R).sum().backward()

    gru = _GRU(sd, p, "cuda")
    st = state.detach().permute(0, 2, 3, 1).contiguous().cuda()
    x6 = inp6.detach().cuda()
    out = torch.empty(B, 4, H, W, 32, device="cuda")
    with autodiff.Tape(x3=False) as tape:
        gru(x6, st, out)
        tape.seed(out, R.permute(0, 1, 3, 4, 2))
        tape.backward()
    torch.cuda.synchronize()
    assert float((out.permute(0, 1, 4, 2, 3).cpu() - fut.detach()).abs().max() / fut.detach().abs().max()) < 1e-4
    worst = {}
    for k, v in leaves.items():
        got = tape.param_grads[k].cpu()
        assert got.shape == v.grad.shape, (k, got.shape, v.grad.shape)
        worst[k] = float((got - v.grad).norm() / v.grad.norm().clamp_min(1e-20))
    worst["d state"] = float((tape.grad(st).permute(0, 3, 1, 2).cpu() - state.grad).norm() / state.grad.norm())
    worst["d inp6"] = float((tape.grad(x6).cpu() - inp6.grad).norm() / inp6.grad.norm())
    print("GRU backward: tensors", len(worst), "worst L2 rel", max(worst.values()))
    bad = {k: e for k, e in worst.items() if e > 1e-3}
    assert len(worst) >= 18 and not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


def test_mlp_building_blocks_backward_match_autograd():
    """The decoder's row-batched building blocks under the tape: concat with broadcast / modulo row mappings -> LayerNorm ->
    Linear+GELU -> Linear (+ residual) -> softplus, and the rot90(flip) copy; parameter and input gradients vs torch autograd."""
    import torch.nn.functional as F
    from thinktwice_amd import _lib, autodiff, layers, ops
    g = torch.Generator().manual_seed(21)
    B = 3
    fflat = torch.randn(B * 4, 64, generator=g, requires_grad=True)
    look = torch.randn(B, 32, generator=g, requires_grad=True)
    temporal = torch.randn(4, 16, generator=g, requires_grad=True)
    sd = {"ln.weight": torch.rand(112, generator=g) + 0.5, "ln.bias": torch.randn(112, generator=g) * 0.1,
          "l1.weight": torch.randn(96, 112, generator=g) * 0.1, "l1.bias": torch.randn(96, generator=g) * 0.1,
          "l2.weight": torch.randn(64, 96, generator=g) * 0.1, "l2.bias": torch.randn(64, generator=g) * 0.1}
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    # ---- torch reference
    hin = torch.cat([fflat.view(B, 4, 64), look.unsqueeze(1).expand(B, 4, 32), temporal.unsqueeze(0).expand(B, 4, 16)], -1)
    hin = hin.reshape(B * 4, 112)
    hn = F.layer_norm(hin, (112,), leaves["ln.weight"], leaves["ln.bias"])
    h = F.gelu(F.linear(hn, leaves["l1.weight"], leaves["l1.bias"]))
    y = F.linear(h, leaves["l2.weight"], leaves["l2.bias"]) + fflat
    z = F.softplus(y)
    R = torch.randn(z.shape, generator=g)
    (z * R).sum().backward()
    # ---- HIP
    l1 = layers.linear_from_sd(sd, "l1", "cuda", act="gelu")
    l2 = layers.linear_from_sd(sd, "l2", "cuda")
    gam, bet = sd["ln.weight"].cuda(), sd["ln.bias"].cuda()
    autodiff.LN_META[gam] = ("ln.weight", "ln.bias")
    ff, lk, tp = fflat.detach().cuda(), look.detach().cuda(), temporal.detach().cuda()
    with autodiff.Tape(x3=False) as tape:
        hin_d = torch.empty(B * 4, 112, device="cuda")
        ops.concat_rows(hin_d, [(ff, 64, 1, 0), (lk, 32, 4, 0), (tp, 16, 1, 4)])
        hn_d = ops.layernorm_rows(hin_d, gam, bet)
        y_d = layers.unrows(l2(l1(layers.rows(hn_d)), res1=layers.rows(ff)))
        z_d = ops.ew(3, y_d, act=_lib.ACT_SOFTPLUS)
        tape.seed(z_d, R)
        tape.backward()
    torch.cuda.synchronize()
    assert float((z_d.cpu() - z.detach()).abs().max()) < 1e-5
    errs = {k: float((tape.param_grads[k].cpu() - v.grad).norm() / v.grad.norm()) for k, v in leaves.items()}
    for name, t, ref in (("d fflat", ff, fflat), ("d look", lk, look), ("d temporal", tp, temporal)):
        errs[name] = float((tape.grad(t).cpu() - ref.grad).norm() / ref.grad.norm())
    print("mlp blocks backward:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < 2e-5, errs
    # rot90(flip) copy
    x = torch.randn(2, 21, 21, 8, generator=g)
    Rr = torch.randn(2, 21, 21, 8, generator=g)
    xd = x.cuda()
    with autodiff.Tape() as tape:
        out = torch.empty(2, 21, 21, 8, device="cuda")
        ops.copy_nhwc(xd, out, rot_flip=True)
        tape.seed(out, Rr)
        tape.backward()
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = torch.rot90(torch.flip(xr, dims=[2]), 1, dims=[2, 3])
    assert float((out.permute(0, 3, 1, 2).cpu() - yr.detach()).abs().max()) == 0.0
    (yr * Rr.permute(0, 3, 1, 2)).sum().backward()
    assert float((tape.grad(xd).permute(0, 3, 1, 2).cpu() - xr.grad).abs().max()) == 0.0


@pytest.mark.parametrize("B", [1, 2])
def test_look_module_backward_matches_oracle_autograd(B):
    """Camera look module of one refinement layer (thinktwice_decoder.py:154-187, multi_scale_deformable_attn_function.py:
    279-344 and :197-277): FPN-side 1x1 projections -> query gather (embeddings, per-sample vectors, bilinear samples of the
    four maps) -> LayerNorm / MLP -> offsets + softmax weights -> deformable sampling of the value projection (with the
    camera / level embedding shift) -> FFN -> slot reduction -> output MLP.  Parameter gradients under the reference names
    and the gradients w.r.t. the FPN features, the measurement vector and the flattened BEV vector vs oracle autograd."""
    from oracle import model_ref as M
    from tests.test_decoder import _inputs
    from thinktwice_amd import autodiff, config, params, weights
    from thinktwice_amd.decoder import ThinkTwiceDecoder
    hw = (128, 256)
    cfg = config.model_config(final_dim=hw)
    sd = params.init_params(cfg, seed=3, parts=("decoder",))
    _, _, fpn, batch, l2i, ida = _inputs(B, hw, seed=4)
    g = torch.Generator().manual_seed(33)
    wp = torch.cumsum(torch.randn(B, 4, 2, generator=g) + torch.tensor([2.0, 0.0]), 1)
    ctrl_sp = torch.rand(B, 4, 4, generator=g) + 0.5
    meas = torch.randn(B, 128, generator=g).requires_grad_(True)
    flat = torch.randn(B, 256, generator=g).requires_grad_(True)
    fpn = [f.clone().requires_grad_(True) for f in fpn]
    p = "decoder.decoder_layers.0.look_module"
    used = (p + ".cam_look_module.", "decoder.fpn_linear", "decoder.cams_embeds", "decoder.level_embeds",
            "decoder.temporal_embedding", "decoder.static_embedding")
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(used)}
    sdr = dict(sd)
    sdr.update(leaves)
    mlvl = [M.conv(sdr, f"decoder.fpn_linear{i}", fpn[i]) for i in range(4)]
    shapes = [tuple(f.shape[2:]) for f in mlvl]
    vals = []
    for lvl, f in enumerate(mlvl):
        v = f.view(B, 4, 256, -1).permute(0, 1, 3, 2)
        vals.append(v + sdr["decoder.cams_embeds"].view(1, 4, 1, 256) + sdr["decoder.level_embeds"][lvl].view(1, 1, 1, 256))
    value_in = torch.cat(vals, 2).reshape(B * 4, -1, 256)
    ref, info = M.look_module(sdr, p, cfg, wp, ctrl_sp, meas, flat, l2i, ida, mlvl, value_in, shapes,
                              sdr["decoder.temporal_embedding"], sdr["decoder.static_embedding"])
    assert info["max_len"] > B, "the test needs queries that project into the cameras"
    R = torch.randn(ref.shape, generator=g)
    (ref * R).sum().backward()

    dec = ThinkTwiceDecoder(config=cfg["cfg"], bev_h=21, bev_w=21).load_state_dict(sd)
    assert dec.fused is None
    lay = dec.layers[0]
    fpn_d = [weights.to_channel_last(f.detach()).cuda() for f in fpn]
    meas_d, flat_d = meas.detach().cuda(), flat.detach().cuda()
    with autodiff.Tape(x3=False) as tape:
        maps = [dec.fpn_linear[i](t) for i, t in enumerate(fpn_d)]
        level_hw = [(m.shape[1], m.shape[2]) for m in maps]
        S = sum(h * w for h, w in level_hw)
        value = dec._project_values_train(lay, maps, B, S)
        out, _ = dec._look(lay, B, wp.cuda(), ctrl_sp.cuda(), meas_d, flat_d, l2i.cuda().float().contiguous(),
                           ida.cuda().float().contiguous(), maps, level_hw, S, (value, 0, None))
        tape.seed(out, R)
        tape.backward()
    torch.cuda.synchronize()
    assert float((out.cpu() - ref.detach()).abs().max() / ref.detach().abs().max()) < 1e-4
    worst, unused = {}, []
    for k, v in leaves.items():
        if v.grad is None:
            unused.append(k)
            continue
        got = tape.param_grads[k].cpu()
        assert got.shape == v.grad.shape, (k, got.shape, v.grad.shape)
        worst[k] = float((got - v.grad).norm() / v.grad.norm().clamp_min(1e-20))
    for i in range(4):
        worst[f"d fpn{i}"] = float((tape.grad(fpn_d[i]).permute(0, 3, 1, 2).cpu() - fpn[i].grad).norm() / fpn[i].grad.norm())
    worst["d meas"] = float((tape.grad(meas_d).cpu() - meas.grad).norm() / meas.grad.norm())
    worst["d flat"] = float((tape.grad(flat_d).cpu() - flat.grad).norm() / flat.grad.norm())
    print("look module backward: tensors", len(worst), "unused", len(unused), "worst L2 rel", max(worst.values()))
    bad = {k: e for k, e in worst.items() if e > 1e-3}
    assert len(worst) >= 30 and not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


@pytest.mark.parametrize("split_k,tol", [(False, 1e-2), (True, 1e-2)], ids=["in-workgroup-K", "product-split-K"])
def test_decoder_backward_matches_oracle_autograd(monkeypatch, split_k, tol):
    """The whole look-and-predict decoder (thinktwice_decoder.py:419-533): coarse heads, five refinement layers (conv-GRU
    + shared flatten network, look module, merge MLP, offset heads, BEV / flattened-feature updates) chained through the
    DETACHED previous outputs (DEC:429-430), and the teacher-forcing pass over the same layers.  Random cotangents on every
    output the training losses read; parameter gradients under the reference names and the gradients handed back to the
    encoder (flattened BEV vector, BEV map, measurement vector, FPN features) vs oracle autograd."""
    from oracle import model_ref as M
    from tests.test_decoder import _inputs
    from thinktwice_amd import autodiff, config, params, synth, weights
    from thinktwice_amd.decoder import ThinkTwiceDecoder
    from thinktwice_amd.encoder_decoder import EncoderDecoder
    from thinktwice_amd.fusion import BEVFusion
    from thinktwice_amd import ops
    # The backward kernels are exact against autograd wherever the forward's ReLU masks agree with the CPU forward (worst
    # 3.0e-4 over the 443 tensors).  WHICH borderline pre-activations flip depends on the forward's K-summation order: with
    # the in-workgroup order none does on this input; the product's ordered cross-workgroup split-K (K >= 2048 layers at
    # M <= 4096: the BEV-update conv, the flatten MLPs) flips a handful on the flatten network's 2 x 2 / 4 x 4 maps and on
    # the 21 x 21 GRU maps (measured worst 6.7e-3 / 2.8e-3).  Both orders are run; both are deterministic.
    # Round 5: the camera / level embedding shift of value_proj (W e, 4 x 256 x 256 per layer) is computed by the library's exact-f32
    # kernel instead of torch.addmm; the 1e-7 difference reaches the same borderline pre-activations through
    # the look features, so the in-workgroup order now shows the SAME flips (6.7e-3 on the flatten network, 2.8e-3 on the GRU
    # biases, every other tensor as before): both arms carry the mask-flip bound.
    monkeypatch.setattr(ops, "_AUTO_SPLITK", split_k)
    B, hw, Rn = 2, (128, 256), 5
    cfg = config.model_config(final_dim=hw)
    sd = params.init_params(cfg, seed=6, parts=("fusion", "decoder"))
    _, _, fpn, batch, l2i, ida = _inputs(B, hw, seed=7)
    tgt = synth.make_train_targets(B, img_hw=hw)
    teacher = {k: tgt[k] for k in ("waypoints", "action_mu", "action_sigma", "future_action_mu", "future_action_sigma")}
    g = torch.Generator().manual_seed(44)
    flat = torch.randn(B, 256, generator=g).requires_grad_(True)
    bev = (torch.randn(B, 32, 21, 21, generator=g) * 0.5).requires_grad_(True)
    meas = torch.randn(B, 128, generator=g).abs().requires_grad_(True)
    fpn = [f.clone().requires_grad_(True) for f in fpn]
    tail = ("MLP10.", "MLP4.", "MLP2.", "conv21_10.", "conv10_4.", "conv4_2.", "output_fc.")
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if (k.startswith("decoder.") or k.startswith(tail)) and v.is_floating_point() and v.dim() > 0
              and not k.endswith(("running_mean", "running_var"))}
    sdr = dict(sd)
    sdr.update(leaves)
    ref = M.decoder_forward(sdr, cfg, flat, bev, meas, l2i, ida, fpn, teacher=teacher)
    ct_ref = torch.cat([torch.cat([ref["mu_branches"], ref["sigma_branches"]], -1).unsqueeze(2),
                        torch.cat([ref["future_mu"], ref["future_sigma"]], -1)], 2)                 # (B, R+1, 4, 4)
    fut_ref = ref["refine_future_BEV_feature"].transpose(1, 2).reshape(B, Rn, 4, 32, 21, 21)        # undo the DEC:481 re-view
    pairs = [("pred_wp", ref["pred_wp"], None), ("ct", ct_ref, None), ("pred_speed", ref["pred_speed"], None),
             ("pred_value_traj", ref["pred_value_traj"], None), ("pred_value_ctrl", ref["pred_value_ctrl"], None),
             ("pred_features_traj", ref["pred_features_traj"], None), ("pred_features_ctrl", ref["pred_features_ctrl"], None),
             ("refine_flattned_BEV_feature", ref["refine_flattned_BEV_feature"], None),
             ("_refine_bev_cl", ref["refine_BEV_feature"], (0, 1, 3, 4, 2)), ("_refine_fut_cl", fut_ref, (0, 1, 2, 4, 5, 3)),
             ("teacher_pred_wp_offset", ref["teacher_pred_wp_offset"], None),
             ("teacher_pred_ctrl_offset_lis", ref["teacher_pred_ctrl_offset_lis"], None),
             ("teacher_refine_flattned_BEV_feature", ref["teacher_refine_flattned_BEV_feature"], None),
             ("_teacher_refine_bev_cl", ref["teacher_refine_BEV_feature"], (0, 1, 3, 4, 2)),
             ("_teacher_fut_cl", ref["teacher_future_BEV_feature"], (0, 1, 2, 4, 5, 3))]
    cot = {name: torch.randn(t.shape, generator=g) for name, t, _ in pairs}
    sum((t * cot[name]).sum() for name, t, _ in pairs).backward()

    dev = torch.device("cuda")
    par = EncoderDecoder.__new__(EncoderDecoder)
    par.device = dev
    par.fusion = BEVFusion(sd, dev)
    dec = ThinkTwiceDecoder(config=cfg["cfg"], bev_h=21, bev_w=21).load_state_dict(sd)
    fpn_d = [weights.to_channel_last(f.detach()).cuda() for f in fpn]
    flat_d, meas_d = flat.detach().cuda(), meas.detach().cuda()
    bev_d = weights.to_channel_last(bev.detach()).cuda()
    t_dev = {k: ([t.cuda() for t in v] if isinstance(v, list) else v.cuda()) for k, v in teacher.items()}
    with autodiff.Tape(x3=False) as tape:
        out = dec(flat_d, bev_d, meas_d, batch["target_point"], par, t_dev, [l2i, ida, [(f, 0, 256) for f in fpn_d], None],
                  channel_last_out=True)
        for name, t, perm in pairs:
            c = cot[name] if perm is None else cot[name].permute(*perm)
            if name == "ct":
                tape.seed(out["mu_branches"], c[:, :, 0, :2])
                tape.seed(out["sigma_branches"], c[:, :, 0, 2:])
                tape.seed(out["future_mu"], c[:, :, 1:, :2])
                tape.seed(out["future_sigma"], c[:, :, 1:, 2:])
            else:
                tape.seed(out[name], c)
        tape.backward()
    torch.cuda.synchronize()
    for name, t, perm in pairs:
        if name == "ct":
            continue
        got = out[name].cpu() if perm is None else out[name].cpu().permute(*[perm.index(i) for i in range(len(perm))])
        assert float((got - t.detach()).abs().max() / t.detach().abs().max()) < 1e-4, name
    worst, unused = {}, []
    for k, v in leaves.items():
        if v.grad is None:
            unused.append(k)
            continue
        assert k in tape.param_grads, k
        got = tape.param_grads[k].cpu()
        assert got.shape == v.grad.shape, (k, got.shape, v.grad.shape)
        worst[k] = float((got - v.grad).norm() / v.grad.norm().clamp_min(1e-20))
    for i in range(4):
        worst[f"d fpn{i}"] = float((tape.grad(fpn_d[i]).permute(0, 3, 1, 2).cpu() - fpn[i].grad).norm() / fpn[i].grad.norm())
    worst["d meas"] = float((tape.grad(meas_d).cpu() - meas.grad).norm() / meas.grad.norm())
    worst["d flat"] = float((tape.grad(flat_d).cpu() - flat.grad).norm() / flat.grad.norm())
    worst["d bev"] = float((tape.grad(bev_d).permute(0, 3, 1, 2).cpu() - bev.grad).norm() / bev.grad.norm())
    print("decoder backward: tensors", len(worst), "unused", len(unused), "worst L2 rel", max(worst.values()))
    # the shared flatten network (BatchNorm folded into the conv epilogue, ReLU on 10x10 / 4x4 / 2x2 maps of 40 images per
    # pass) sees the mask flips described above the whole-encoder test: its own bound
    w_tail = max(e for k, e in worst.items() if k.startswith(tail))
    rest = sorted(((k, e) for k, e in worst.items() if not k.startswith(tail)), key=lambda kv: -kv[1])
    print("  flatten network worst", w_tail, "| other tensors worst 5:", rest[:5])
    bad = {k: e for k, e in worst.items() if e > tol}
    assert len(worst) >= 400 and not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
