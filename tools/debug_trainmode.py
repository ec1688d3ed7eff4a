"""Debug aid (GPU box): HIP forward_train in train mode vs the oracle in train mode, every prediction tensor + mids."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from oracle import model_ref as M, train_ref as TR          # noqa: E402
from thinktwice_amd import layers, model as tm, ops, params, synth  # noqa: E402
import test_train_step as T                                  # noqa: E402

B, hw, npts, rng = 2, (128, 256), 20000, 20240607
m, cfg = tm.build_thinktwice(final_dim=hw, dtype=torch.float32)
sd = params.init_params(cfg, seed=0)
batch = synth.make_batch(B, img_hw=hw, num_points=npts, jitter_calib=77)
batch.update(synth.make_train_targets(B, img_hw=hw))
with torch.no_grad(), TR.train_mode():
    torch.manual_seed(rng)
    cam = M.lss_forward(sd, "img_encoder", cfg, batch["img"], batch["img_metas"])
    cam_bev = M.rot_flip(cam["bev"])
    meas = M.measurement_feat(sd, batch)
    lid = [M.rot_flip(t) for t in M.lidar_net(sd, "lidar_encoder", cfg, batch["points"][:, -1])]
    flat, bev32, mids = M.fusion(sd, cam_bev, lid[0])
    teacher = {k: batch[k] for k in ("waypoints", "action_sigma", "action_mu", "future_action_sigma", "future_action_mu")}
    pred = M.decoder_forward(sd, cfg, flat, bev32, meas, cam["lidar2img"], cam["ida_mat"], cam["fpn_feats"], teacher=teacher)
m.load_state_dict(sd)
m.train()
ops.DROPOUT_MASKS = iter([T._reference_dropout_masks(B, cfg, hw, rng)])
saved, layers.BN_TRAIN = layers.BN_TRAIN, True
try:
    from thinktwice_amd import losses as LS
    teacher_d = {k: batch[k] for k in LS.TEACHER_KEYS}
    teacher_d = {k: ([t.cuda() for t in v] if isinstance(v, (list, tuple)) else v.cuda()) for k, v in teacher_d.items()}
    got = m.forward_inference(batch, teacher=teacher_d, channel_last_out=False)
finally:
    layers.BN_TRAIN = saved
torch.cuda.synchronize()


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a.reshape(b.shape) - b).abs().max() / b.abs().max().clamp_min(1e-9))


print("keys", sorted(k for k in got if not k.startswith("_"))[:40])
for k, v in pred.items():
    if torch.is_tensor(v) and k in got and torch.is_tensor(got[k]):
        print(f"{k:32s} {tuple(v.shape)} rel err {rel(got[k], v):.3e}")
for name, ref in (("_flat", flat), ("_meas", meas)):
    if name in got:
        print(name, rel(got[name], ref))
for i, (a, b) in enumerate(zip(got.get("_mid_bev_cl", []), mids)):
    if a is not None and b is not None:
        print("mid", i, rel(a.permute(0, 3, 1, 2), b))
print("cam bev (rot-flipped)", rel(got["_cam_bev_cl"].permute(0, 3, 1, 2), cam_bev))
print("lidar bev", rel(got["_lidar_bev_cl"].permute(0, 3, 1, 2), lid[0]))
ops.DROPOUT_MASKS = iter([T._reference_dropout_masks(B, cfg, hw, rng)])
layers.BN_TRAIN = True
m.img_encoder._dbg = {}
try:
    enc = m.img_encoder(batch["img"].cuda(), batch["img_metas"], channel_last=True)
finally:
    layers.BN_TRAIN = False
    ops.DROPOUT_MASKS = None
torch.cuda.synchronize()
for i, ((t, off, c), ref) in enumerate(zip(enc["_fpn_cl"], cam["fpn_feats"])):
    print("fpn", i, rel(t[..., off:off + c].permute(0, 3, 1, 2), ref))
print("seg", rel(enc["_seg_cl"][..., :12].permute(0, 3, 1, 2), cam["seg"]))
print("depth logits/prob?", tuple(enc["_depth_cl"].shape), tuple(cam["depth"].shape))
d_h = torch.softmax(enc["_depth_cl"].float(), -1).permute(0, 3, 1, 2)
print("depth (softmax of HIP logits vs oracle)", rel(d_h, cam["depth"]), "raw", rel(enc["_depth_cl"].permute(0, 3, 1, 2), cam["depth"]))
print("bev", rel(enc["_bev_cl"].permute(0, 3, 1, 2), cam["bev"]))
print("key bev", tuple(enc["_key_bev_cl"].shape))

# ---- DepthNet stage by stage: the oracle run on the HIP path's own neck output (key sweep = first B*N images)
import torch.nn.functional as F   # noqa: E402
dbg = m.img_encoder._dbg
BN_ = B * 4
nchw = lambda t: t.detach().float().cpu().permute(0, 3, 1, 2).contiguous()   # noqa: E731
p = "img_encoder.depth_net"
from thinktwice_amd import camera  # noqa: E402
mats = camera.stack_img_metas(batch["img_metas"], 4)
mlp_in = camera.depth_mlp_input(mats)
with torch.no_grad(), TR.train_mode():
    mm = M.bn(sd, p + ".bn", mlp_in)
    print("bn22", rel(dbg["m24"][:, :mm.shape[1]], mm), "max", float(mm.abs().max()))
    x_in = nchw(dbg["src"][:BN_])
    xr = F.relu(M.bn(sd, p + ".reduce_conv.1", M.conv(sd, p + ".reduce_conv.0", x_in, 1, 1)))
    print("reduce", rel(nchw(dbg["reduce"][:BN_]), xr))

    def se(name, feat):
        v = M.linear(sd, f"{p}.{name}_mlp.fc2", F.relu(M.linear(sd, f"{p}.{name}_mlp.fc1", mm)))[..., None, None]
        v = M.conv(sd, f"{p}.{name}_se.conv_expand", F.relu(M.conv(sd, f"{p}.{name}_se.conv_reduce", v)))
        return feat * torch.sigmoid(v)
    d = se("depth", nchw(dbg["reduce"][:BN_]))
    print("se_depth", rel(nchw(dbg["se_depth"][:BN_]), d))
    d = nchw(dbg["se_depth"][:BN_])
    for i in range(3):
        d = M.basic_block(sd, f"{p}.depth_conv.{i}", d)
    print("blocks", rel(nchw(dbg["blocks"][:BN_]), d))
    a = p + ".depth_conv.3"
    x = nchw(dbg["blocks"][:BN_])
    x5 = F.adaptive_avg_pool2d(x, (1, 1))
    x5 = F.relu(M.bn(sd, a + ".global_avg_pool.2", M.conv(sd, a + ".global_avg_pool.1", x5)))
    print("x5", rel(nchw(dbg["x5"][:BN_]), x5))
    br = F.relu(M.bn(sd, a + ".aspp2.bn", M.conv(sd, a + ".aspp2.atrous_conv", x, 1, 6, 6)))
    mid = br.shape[1]
    print("aspp2", rel(nchw(dbg["aspp_cat"][:BN_])[:, mid:2 * mid], br))
    old, M.TRAIN_MODE = M.TRAIN_MODE, True
    torch.manual_seed(rng)
    y = M.aspp(sd, a, x)
    keep = (y != 0)
    pre = nchw(dbg["aspp_pre_dropout"][:BN_])
    print("aspp_out (kept elements, x2 dropout scale)", float(((pre * 2.0 - y)[keep]).abs().max() / y.abs().max()))
