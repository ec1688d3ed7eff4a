"""CPU restatement of the reference TRAINING forward (losses), test infrastructure only (see oracle/__init__.py).

`forward_train(sd, cfg, batch)` follows `EncoderDecoder.forward_train` (encoder_decoder_framework.py:147-191):
the inference forward (oracle/model_ref.py), the decoder's teacher-forcing pass (thinktwice_decoder.py:491-533),
`ThinkTwiceDecoder.loss` (thinktwice_decoder.py:536-619), the focal segmentation loss (utils.py:31-47,
encoder_decoder_framework.py:172-176) and the depth BCE (encoder_decoder_framework.py:179-190, 441-481).

Two modes.  Default: BatchNorm layers use their RUNNING statistics (the in-repo reference code has no `self.training`
branch, so this is exactly what the reference computes under `model.eval()`, and what a frozen-BN fine-tune runs);
pinned by tests/golden/f10_train_losses_b2.npz (gen_golden.py F10).  Inside `with train_mode():` every BatchNorm
normalises with batch statistics (per sweep: the reference runs each sweep through the camera trunk separately,
lss.py:690-717) and the ASPP Dropout(0.5) (lss.py:91) is live, drawing from torch's global RNG -- what `model.train()`
computes; pinned by f11_train_losses_trainmode_b2.npz (F11, same seed before both forwards).
"""
import contextlib
import torch
import torch.nn.functional as F
from torch.distributions import Beta, kl_divergence

from . import model_ref as M

DISTIL_INDEX = (2, 3, 4, 5)                                   # DEC:284
DISTIL_W = {2: 0.25, 3: 1.0 / 3.0, 4: 1.0 / 4.0, 5: 1.0 / 11.0}   # DEC:285
WP_W = ACTION_W = 15.0                                        # DEC:286-287


@contextlib.contextmanager
def train_mode():
    """model.train() semantics for the oracle (batch-statistics BN, live ASPP dropout)."""
    old = M.TRAIN_MODE
    M.TRAIN_MODE = True
    try:
        yield
    finally:
        M.TRAIN_MODE = old


def action_beta(alpha, beta):
    """ThinkTwiceDecoder._get_action_beta DEC:622-637 (mode of the Beta with the edge cases), mapped to [-1, 1]."""
    x = torch.zeros_like(alpha)
    x[:, 1] += 0.5
    m1 = (alpha > 1) & (beta > 1)
    x[m1] = (alpha[m1] - 1) / (alpha[m1] + beta[m1] - 2)
    x[(alpha <= 1) & (beta > 1)] = 0.0
    x[(alpha > 1) & (beta <= 1)] = 1.0
    m4 = (alpha <= 1) & (beta <= 1)
    x[m4] = alpha[m4] / torch.clamp(alpha[m4] + beta[m4], min=1e-5)
    return x * 2 - 1


def _clamped_sl1(pred, gt):
    return torch.clamp(F.smooth_l1_loss(pred, gt, reduction="none"), min=-5.0, max=5.0).mean()


def decoder_loss(cfg, batch, pred, mid_bev):
    """ThinkTwiceDecoder.loss DEC:536-619.  `mid_bev`: [32x21x21, 64x10x10, 128x4x4, 256x2x2] encoder maps; the
    reference indexes `mid_BEV_feature[2..5]` of a 4-element list... it is built with two leading entries
    (EDF:212-233: lidar map and fused map first), so index 2 is the 32x21x21 map."""
    c = cfg["cfg"]
    L = {}
    gt_speed = batch["speed"].float().view(-1, 1) / 12.0
    gt_value = batch["value"].view(-1, 1)
    gt_feat = batch["feature"]
    gt_wp = batch["waypoints"]
    with torch.no_grad():
        l1a = F.l1_loss(action_beta(pred["mu_branches"][:, -1], pred["sigma_branches"][:, -1]),
                        action_beta(batch["action_mu"], batch["action_sigma"]), reduction="none").mean(0)
        L["current_throttle_brake_offset"], L["current_steer_offset"] = l1a[0], l1a[1]
        off = F.l1_loss(pred["pred_wp"][:, -1], gt_wp, reduction="none").mean(0).mean(0)
        L["longitudinal_offset"], L["lateral_offset"] = off[0], off[1]
    kl = kl_divergence(Beta(batch["action_mu"].unsqueeze(1), batch["action_sigma"].unsqueeze(1)),
                       Beta(pred["mu_branches"], pred["sigma_branches"]))
    L["action_loss"] = kl.mean() * ACTION_W
    L["speed_loss"] = F.smooth_l1_loss(pred["pred_speed"], gt_speed)
    L["value_loss"] = (F.smooth_l1_loss(pred["pred_value_traj"], gt_value) +
                       F.smooth_l1_loss(pred["pred_value_ctrl"], gt_value, reduction="none")) * c["value_weight"]
    L["flattened_feature_loss"] = (F.smooth_l1_loss(pred["pred_features_traj"], gt_feat) +
                                   F.smooth_l1_loss(pred["pred_features_ctrl"], gt_feat)) * c["features_weight"]
    fmu = torch.stack(batch["future_action_mu"][:-1], 1).unsqueeze(1)
    fsg = torch.stack(batch["future_action_sigma"][:-1], 1).unsqueeze(1)
    L["future_action_loss"] = kl_divergence(Beta(fmu, fsg), Beta(pred["future_mu"], pred["future_sigma"])).mean() \
        * ACTION_W * 0.25
    R = pred["pred_wp"].shape[1]
    L["wp_loss"] = F.smooth_l1_loss(pred["pred_wp"], gt_wp.unsqueeze(1).repeat(1, R, 1, 1)) * WP_W
    for i in DISTIL_INDEX:
        L[f"BEV_feature_loss{i}"] = _clamped_sl1(mid_bev[i], batch["grid_feature"][i]) * DISTIL_W[i]
    rb = pred["refine_BEV_feature"]
    g2 = batch["grid_feature"][2].unsqueeze(1).repeat(1, rb.shape[1], 1, 1, 1)
    L["refine_BEV_feature_loss2"] = _clamped_sl1(rb, g2) * DISTIL_W[2]
    rf = pred["refine_flattned_BEV_feature"]
    L["refine_flattened_feature_loss"] = _clamped_sl1(rf, gt_feat.unsqueeze(1).repeat(1, rf.shape[1], 1)) \
        * c["features_weight"] * 0.1
    L["teacher_wp_loss"] = F.smooth_l1_loss(pred["teacher_pred_wp_offset"],
                                            torch.zeros_like(pred["teacher_pred_wp_offset"]))
    L["teacher_action_loss"] = F.smooth_l1_loss(pred["teacher_pred_ctrl_offset_lis"],
                                                torch.zeros_like(pred["teacher_pred_ctrl_offset_lis"]))
    gfut = torch.stack([g[2] for g in batch["future_grid_feature"]], 1)
    pf = pred["teacher_future_BEV_feature"]                      # (N, R, T, C, W, H)
    L["teacher_future_BEV_feature_loss2"] = _clamped_sl1(pf, gfut.unsqueeze(1).repeat(1, pf.shape[1], 1, 1, 1, 1)) \
        * DISTIL_W[2]
    tb = pred["teacher_refine_BEV_feature"]
    L["teacher_refine_BEV_feature_loss2"] = _clamped_sl1(
        tb, batch["grid_feature"][2].unsqueeze(1).repeat(1, tb.shape[1], 1, 1, 1)) * DISTIL_W[2]
    tf = pred["teacher_refine_flattned_BEV_feature"]
    L["teacher_refine_flattened_feature_loss"] = _clamped_sl1(tf, gt_feat.unsqueeze(1).repeat(1, tf.shape[1], 1)) \
        * c["features_weight"]
    return L


def seg_loss(seg_pred, gt_seg, factor=2):
    """EDF:172-176 + get_downsampled_gt_seg EDF:483-489 ([3P] torchvision nearest Resize = F.interpolate nearest)
    + FocalLoss utils.py:31-47 (alpha 0.5, gamma 2, on the MEAN cross entropy)."""
    B, N, H, W = gt_seg.shape
    gt = F.interpolate(gt_seg.view(1, B * N, H, W).float(), size=(H // factor, W // factor), mode="nearest")[0].long()
    logpt = -F.cross_entropy(seg_pred.float(), gt, ignore_index=255)
    pt = torch.exp(logpt)
    return -((1 - pt) ** 2) * 0.5 * logpt * 10


def depth_loss(depth_logits, gt_depth, d_bound, factor=16):
    """EDF:179-190 + get_downsampled_gt_depth EDF:441-481 (min-pooling over factor x factor cells, 0 = no return,
    one-hot depth bin, BCE with logits on the foreground cells)."""
    B, N, H, W = gt_depth.shape
    D = int((d_bound[1] - d_bound[0]) / d_bound[2])
    g = gt_depth.view(B * N, H // factor, factor, W // factor, factor, 1).permute(0, 1, 3, 5, 2, 4).contiguous()
    g = g.view(-1, factor * factor)
    g = torch.min(torch.where(g == 0.0, 1e5 * torch.ones_like(g), g), dim=-1).values
    g = g.view(B * N, H // factor, W // factor)
    g = (g - (d_bound[0] - d_bound[2])) / d_bound[2]
    g = torch.where((g < D + 1) & (g >= 0.0), g, torch.zeros_like(g))
    onehot = F.one_hot(g.long(), num_classes=D + 1).view(-1, D + 1)[:, 1:].float()
    pred = depth_logits.permute(0, 2, 3, 1).contiguous().view(-1, D)
    fg = torch.max(onehot, dim=1).values > 0.0
    return F.binary_cross_entropy_with_logits(pred[fg], onehot[fg], reduction="none").sum() / max(1.0, fg.sum())


def total_loss(losses):
    """EncoderDecoder._parse_losses EDF:409-439: mean of every entry, sum of those whose name contains 'loss'."""
    return sum(v.mean() for k, v in losses.items() if "loss" in k)


def forward_train(sd, cfg, batch):
    """EncoderDecoder.forward_train EDF:147-191 with running-statistics BN (see module docstring)."""
    cam = M.lss_forward(sd, "img_encoder", cfg, batch["img"], batch["img_metas"])
    cam_bev = M.rot_flip(cam["bev"])
    meas = M.measurement_feat(sd, batch)
    lid = [M.rot_flip(t) for t in M.lidar_net(sd, "lidar_encoder", cfg, batch["points"][:, -1])]
    flat, bev32, mids = M.fusion(sd, cam_bev, lid[0])
    teacher = {k: batch[k] for k in ("waypoints", "action_sigma", "action_mu", "future_action_sigma",
                                     "future_action_mu")}
    pred = M.decoder_forward(sd, cfg, flat, bev32, meas, cam["lidar2img"], cam["ida_mat"], cam["fpn_feats"],
                             teacher=teacher)
    losses = decoder_loss(cfg, batch, pred, mids)
    if cfg["cfg"].get("use_seg"):
        losses["seg_loss"] = seg_loss(cam["seg"], batch["seg"])
    if cfg["cfg"].get("use_depth"):
        losses["depth_loss"] = depth_loss(cam["depth"], batch["depth"], cfg["img_encoder"]["d_bound"],
                                          cfg["img_encoder"]["downsample_factor"])
    return losses, pred
