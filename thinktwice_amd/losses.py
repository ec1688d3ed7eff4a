"""Loss half of the training forward on the device reductions of csrc/losses.hip.

`decoder_loss` mirrors ThinkTwiceDecoder.loss (thinktwice_decoder.py:536-619) term by term and in the reference's key
order; `seg_loss` / `depth_loss` mirror encoder_decoder_framework.py:172-190 (+ utils.py:31-47, :441-489);
`parse_losses` is EncoderDecoder._parse_losses (encoder_decoder_framework.py:409-439).  Every term is ONE launch of a
tt_loss_* reduction writing a device float; nothing here synchronises with the host except `parse_losses`' `.item()`
calls, which the reference makes too.
"""
import ctypes
from collections import OrderedDict

import torch

from . import autodiff, ops
from ._lib import check, cur_stream, lib, ptr, require_cuda

F32 = torch.float32
DISTIL_INDEX = (2, 3, 4, 5)                                            # DEC:284
DISTIL_W = {2: 0.25, 3: 1.0 / 3.0, 4: 1.0 / 4.0, 5: 1.0 / 11.0}       # DEC:285
WP_W = ACTION_W = 15.0                                                 # DEC:286-287

_ll, _c, _f = ctypes.c_longlong, ctypes.c_int, ctypes.c_float


class LossReducer:
    """Owns the (zeroed once) reduction workspace and the output scalars of one device."""

    def __init__(self, device):
        L = lib()
        L.tt_loss_workspace_bytes.restype = ctypes.c_longlong
        self.device = torch.device(device)
        self.ws = torch.zeros(int(L.tt_loss_workspace_bytes()), dtype=torch.uint8, device=self.device)

    def _f(self, t):
        t = t.to(self.device, F32)
        return t if t.is_contiguous() else t.contiguous()

    def _out(self, n=1):
        return torch.empty(n, dtype=F32, device=self.device)

    def smooth_l1(self, pred, target=None, clamp_max=0.0, scale=1.0, reduce=True):
        """pred (N, [R,] ...) vs target (N, ...) broadcast over R (None: zeros)."""
        pred_in = pred
        pred = self._f(pred)
        require_cuda(pred)
        N = pred.shape[0]
        if target is None:
            rep, inner = 1, pred.numel() // N
        else:
            target = self._f(target)
            inner = target.numel() // N
            rep = pred.numel() // (N * inner)
            assert target.shape[0] == N and rep * inner * N == pred.numel(), (pred.shape, target.shape)
        out = self._out(1) if reduce else torch.empty_like(pred)
        check(lib().tt_loss_smooth_l1(ptr(pred), ptr(target), _ll(N), _c(rep), _ll(inner), _f(clamp_max), _f(scale),
                                      _c(1 if reduce else 0), ptr(out), ptr(self.ws), cur_stream(self.device)),
              "tt_loss_smooth_l1")
        tape = autodiff.TAPE
        if tape is not None:
            # the term enters the total as its MEAN with weight 1 (parse_losses; an unreduced term is averaged there)
            def bwd():
                d = torch.empty_like(pred)
                check(lib().tt_loss_smooth_l1_bwd(ptr(pred), ptr(target), _ll(N), _c(rep), _ll(inner), _f(clamp_max),
                                                  _f(scale), ptr(d), cur_stream(self.device)), "tt_loss_smooth_l1_bwd")
                tape.grad(pred_in).add_(d.view(pred_in.shape))
            tape.nodes.append(bwd)
        return out[0] if reduce else out

    def beta_kl(self, t_alpha, t_beta, p_alpha, p_beta, scale=1.0):
        """mean KL(Beta(target) || Beta(pred)); target (N, ...) broadcast over pred (N, R, ...)."""
        pa_in, pb_in = p_alpha, p_beta
        t_alpha, t_beta, p_alpha, p_beta = (self._f(t) for t in (t_alpha, t_beta, p_alpha, p_beta))
        N = p_alpha.shape[0]
        inner = t_alpha.numel() // N
        rep = p_alpha.numel() // (N * inner)
        assert rep * inner * N == p_alpha.numel() and p_beta.shape == p_alpha.shape and t_beta.shape == t_alpha.shape
        out = self._out()
        check(lib().tt_loss_beta_kl(ptr(t_alpha), ptr(t_beta), ptr(p_alpha), ptr(p_beta), _ll(N), _c(rep), _ll(inner),
                                    _f(scale), ptr(out), ptr(self.ws), cur_stream(self.device)), "tt_loss_beta_kl")
        tape = autodiff.TAPE
        if tape is not None:
            def bwd():
                da, db = torch.empty_like(p_alpha), torch.empty_like(p_beta)
                check(lib().tt_loss_beta_kl_bwd(ptr(t_alpha), ptr(t_beta), ptr(p_alpha), ptr(p_beta), _ll(N), _c(rep),
                                                _ll(inner), _f(scale), ptr(da), ptr(db), cur_stream(self.device)),
                      "tt_loss_beta_kl_bwd")
                tape.grad(pa_in).add_(da.view(pa_in.shape))
                tape.grad(pb_in).add_(db.view(pb_in.shape))
            tape.nodes.append(bwd)
        return out[0]

    def l1_cols(self, pred, target, pred_beta=None, target_beta=None):
        """column means of |pred - target| over all leading dims; with *_beta: of the Beta modes (DEC:622-637)."""
        pred, target = self._f(pred), self._f(target)
        cols = pred.shape[-1]
        rows = pred.numel() // cols
        assert target.shape == pred.shape
        pb = None if pred_beta is None else self._f(pred_beta)
        tb = None if target_beta is None else self._f(target_beta)
        out = self._out(cols)
        check(lib().tt_loss_l1_cols(ptr(pred), ptr(pb), _ll(cols), ptr(target), ptr(tb), _ll(rows), _c(cols), ptr(out),
                                    ptr(self.ws), cur_stream(self.device)), "tt_loss_l1_cols")
        return out

    def seg_focal(self, logits_cl, labels, num_classes=12, factor=2):
        """logits_cl (B*N, H/f, W/f, >= num_classes) channel-last; labels (B, N, H, W) class ids (255 = ignore)."""
        labels = self._f(labels)
        B, N, H, W = labels.shape
        assert logits_cl.dtype == F32 and logits_cl.is_contiguous()
        assert tuple(logits_cl.shape[:3]) == (B * N, H // factor, W // factor), (logits_cl.shape, labels.shape)
        out = self._out(3)              # loss, mean cross entropy, contributing pixels (the last two feed the backward)
        check(lib().tt_loss_seg_focal(ptr(logits_cl), _c(logits_cl.shape[-1]), _c(num_classes), ptr(labels), _c(B * N),
                                      _c(H), _c(W), _c(factor), ptr(out), ptr(self.ws), cur_stream(self.device)),
              "tt_loss_seg_focal")
        self.seg_aux = out[1:]
        return out[0]

    def seg_focal_bwd(self, logits_cl, labels, num_classes=12, factor=2, upstream=None):
        """d(seg loss)/d(logits) (channel-last, padding channels 0) for the logits / labels of the last seg_focal call;
        `upstream`: device scalar d(total loss)/d(seg loss) or None (= 1)."""
        labels = self._f(labels)
        B, N, H, W = labels.shape
        d = torch.empty_like(logits_cl)
        check(lib().tt_loss_seg_focal_bwd(ptr(logits_cl), _c(logits_cl.shape[-1]), _c(num_classes), ptr(labels), _c(B * N),
                                          _c(H), _c(W), _c(factor), ptr(self.seg_aux), ptr(upstream), ptr(d),
                                          cur_stream(self.device)), "tt_loss_seg_focal_bwd")
        return d

    def depth_bce(self, logits_cl, gt_depth, d_bound, factor=16):
        """logits_cl (B*N, H/f, W/f, >= D) channel-last depth logits; gt_depth (B, N, H, W) metres, 0 = no return."""
        gt_depth = self._f(gt_depth)
        B, N, H, W = gt_depth.shape
        D = int((d_bound[1] - d_bound[0]) / d_bound[2])
        assert logits_cl.dtype == F32 and logits_cl.is_contiguous()
        assert tuple(logits_cl.shape[:3]) == (B * N, H // factor, W // factor) and logits_cl.shape[-1] >= D
        out = self._out(2)              # loss, divisor max(1, #foreground cells) (feeds the backward)
        check(lib().tt_loss_depth_bce(ptr(logits_cl), _c(logits_cl.shape[-1]), _c(D), ptr(gt_depth), _c(B * N), _c(H),
                                      _c(W), _c(factor), _f(d_bound[0]), _f(d_bound[2]), ptr(out), ptr(self.ws),
                                      cur_stream(self.device)), "tt_loss_depth_bce")
        self.depth_aux = out[1:]
        return out[0]

    def depth_bce_bwd(self, logits_cl, gt_depth, d_bound, factor=16, upstream=None):
        """d(depth loss)/d(logits) for the logits / depth maps of the last depth_bce call."""
        gt_depth = self._f(gt_depth)
        B, N, H, W = gt_depth.shape
        D = int((d_bound[1] - d_bound[0]) / d_bound[2])
        d = torch.empty_like(logits_cl)
        check(lib().tt_loss_depth_bce_bwd(ptr(logits_cl), _c(logits_cl.shape[-1]), _c(D), ptr(gt_depth), _c(B * N), _c(H),
                                          _c(W), _c(factor), _f(d_bound[0]), _f(d_bound[2]), ptr(self.depth_aux),
                                          ptr(upstream), ptr(d), cur_stream(self.device)), "tt_loss_depth_bce_bwd")
        return d


def decoder_loss(red, c, batch, pred, mid_bev, channel_last=False):
    """ThinkTwiceDecoder.loss DEC:536-619.  `pred`: forward_inference(batch, teacher=...) outputs in the reference's
    layouts (channel_last_out=False); `mid_bev[2..5]`: the encoder's 32x21x21 ... 256x2x2 maps, NCHW; `c`: the model's
    train_cfg (value_weight, features_weight).  `channel_last`: the BEV-map terms read the forward's own channel-last
    tensors (`pred` from channel_last_out=True, `mid_bev` channel-last) against targets permuted to match -- the terms are
    elementwise means, so the values are the same, and under the training tape the gradients land on the tensors the
    forward produced."""
    dev = red.device

    def cl(t, nd=4):        # target (..., C, H, W) -> (..., H, W, C)
        if not channel_last:
            return t
        t = t.to(dev, F32)
        return t.permute(*range(t.dim() - 3), t.dim() - 2, t.dim() - 1, t.dim() - 3).contiguous()

    k_rbev, k_tfut, k_trbev = (("_refine_bev_cl", "_teacher_fut_cl", "_teacher_refine_bev_cl") if channel_last else
                               ("refine_BEV_feature", "teacher_future_BEV_feature", "teacher_refine_BEV_feature"))
    L = OrderedDict()
    gt_speed = batch["speed"].to(dev, F32).view(-1, 1) * (1.0 / 12.0)
    gt_value = batch["value"].to(dev, F32).view(-1, 1)
    gt_feat, gt_wp = batch["feature"], batch["waypoints"]
    a_mu, a_sg = batch["action_mu"], batch["action_sigma"]
    off = red.l1_cols(pred["mu_branches"][:, -1], a_mu, pred["sigma_branches"][:, -1], a_sg)       # DEC:544-547
    L["current_throttle_brake_offset"], L["current_steer_offset"] = off[0], off[1]
    off = red.l1_cols(pred["pred_wp"][:, -1], gt_wp)                                                # DEC:548-551
    L["longitudinal_offset"], L["lateral_offset"] = off[0], off[1]
    L["action_loss"] = red.beta_kl(a_mu, a_sg, pred["mu_branches"], pred["sigma_branches"], ACTION_W)
    L["speed_loss"] = red.smooth_l1(pred["pred_speed"], gt_speed)
    # (scalar mean over the trajectory head + per-sample terms of the control head) * weight: shape (B, 1), DEC:559-561
    L["value_loss"] = red.smooth_l1(pred["pred_value_ctrl"], gt_value, scale=c["value_weight"], reduce=False) + \
        red.smooth_l1(pred["pred_value_traj"], gt_value, scale=c["value_weight"])
    L["flattened_feature_loss"] = red.smooth_l1(pred["pred_features_traj"], gt_feat, scale=c["features_weight"]) + \
        red.smooth_l1(pred["pred_features_ctrl"], gt_feat, scale=c["features_weight"])
    fmu = torch.stack([t.to(dev, F32) for t in batch["future_action_mu"][:-1]], 1)                  # (N, T-1, 2)
    fsg = torch.stack([t.to(dev, F32) for t in batch["future_action_sigma"][:-1]], 1)
    L["future_action_loss"] = red.beta_kl(fmu, fsg, pred["future_mu"], pred["future_sigma"], ACTION_W * 0.25)
    L["wp_loss"] = red.smooth_l1(pred["pred_wp"], gt_wp, scale=WP_W)
    for i in DISTIL_INDEX:
        L[f"BEV_feature_loss{i}"] = red.smooth_l1(mid_bev[i], cl(batch["grid_feature"][i]), clamp_max=5.0,
                                                  scale=DISTIL_W[i])
    g2 = cl(batch["grid_feature"][2])
    L["refine_BEV_feature_loss2"] = red.smooth_l1(pred[k_rbev], g2, clamp_max=5.0, scale=DISTIL_W[2])
    L["refine_flattened_feature_loss"] = red.smooth_l1(pred["refine_flattned_BEV_feature"], gt_feat, clamp_max=5.0,
                                                       scale=c["features_weight"] * 0.1)
    L["teacher_wp_loss"] = red.smooth_l1(pred["teacher_pred_wp_offset"])
    L["teacher_action_loss"] = red.smooth_l1(pred["teacher_pred_ctrl_offset_lis"])
    gfut = cl(torch.stack([g[2].to(dev, F32) for g in batch["future_grid_feature"]], 1))           # (N, T, C, W, H)
    L["teacher_future_BEV_feature_loss2"] = red.smooth_l1(pred[k_tfut], gfut, clamp_max=5.0, scale=DISTIL_W[2])
    L["teacher_refine_BEV_feature_loss2"] = red.smooth_l1(pred[k_trbev], g2, clamp_max=5.0, scale=DISTIL_W[2])
    L["teacher_refine_flattened_feature_loss"] = red.smooth_l1(pred["teacher_refine_flattned_BEV_feature"], gt_feat,
                                                               clamp_max=5.0, scale=c["features_weight"])
    return L


def parse_losses(losses):
    """EncoderDecoder._parse_losses EDF:409-439: log_vars = mean of every entry, loss = sum of the entries whose name
    contains 'loss'; under torch.distributed the logged values are averaged over the ranks -- the returned `loss` tensor
    stays local.  The reference issues one all-reduce and one `.item()` per logged value (~24 latency-bound collectives
    and host syncs per iteration); here the values are packed into ONE vector: one all-reduce over xGMI, one
    device-to-host copy."""
    import torch.distributed as dist
    log_vars = OrderedDict()
    for name, value in losses.items():
        if torch.is_tensor(value):
            log_vars[name] = value.mean()
        elif isinstance(value, list):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError(f"{name} is not a tensor or list of tensors")
    loss = sum(v for k, v in log_vars.items() if "loss" in k)
    log_vars["loss"] = loss
    packed = torch.stack([v.detach().to(torch.float32).reshape(()) for v in log_vars.values()])
    if dist.is_available() and dist.is_initialized():
        if packed.is_cuda and dist.get_backend() == "gloo":           # test rigs without RCCL
            packed = packed.cpu()
        dist.all_reduce(packed.div_(dist.get_world_size()))
    for name, value in zip(list(log_vars), packed.tolist()):
        log_vars[name] = value
    return loss, log_vars


TEACHER_KEYS = ("waypoints", "action_sigma", "action_mu", "future_action_sigma", "future_action_mu")
