import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import model_ref as M, train_ref as TR
from thinktwice_amd import autodiff, config, layers, params, weights
from thinktwice_amd.lss import LSS
hw, NI = (64, 128), 4
cfg = config.model_config(final_dim=hw)
sd = params.init_params(cfg, seed=3, parts=("img_encoder",))
leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()
          if (k.startswith("img_encoder.img_backbone") or k.startswith("img_encoder.img_neck"))
          and v.is_floating_point() and not k.endswith(("running_mean", "running_var"))}
sdr = dict(sd); sdr.update(leaves)
g = torch.Generator().manual_seed(7)
img = torch.randn(NI, 3, *hw, generator=g)
with TR.train_mode():
    outs = M.pafpn(sdr, "img_encoder.img_neck", M.resnet50(sdr, "img_encoder.img_backbone", img))
    R = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * r).sum() for o, r in zip(outs, R)).backward()
    p = "img_encoder.img_backbone"
    with torch.no_grad():
        x0 = F.max_pool2d(F.relu(M.bn(sd, p + ".bn1", M.conv(sd, p + ".conv1", img, 2, 3))), 3, 2, 1)
        z = M.conv(sd, p + ".layer1.0.conv1", x0)
        var = z.var(dim=(0, 2, 3), unbiased=False)
        xvar = x0.var(dim=(0, 2, 3), unbiased=False)
enc_cfg = {k: v for k, v in cfg["img_encoder"].items() if k != "type"}
enc = LSS(**enc_cfg, dtype=torch.float32).load_state_dict(sd)
x = weights.to_channel_last(img, torch.float32).cuda()
layers.BN_TRAIN = True
with autodiff.Tape(x3=False) as tape:
    bufs = enc._trunk(x)
    for (t, off, c), r in zip(enc._fpn_views(bufs), R):
        tape.seed(t[..., off:off + c], r.permute(0, 2, 3, 1))
    tape.backward()
layers.BN_TRAIN = False
torch.cuda.synchronize()
k = p + ".layer1.0.conv1.weight"
got, want = tape.param_grads[k].cpu(), leaves[k].grad
err = (got - want).abs().amax(dim=(1, 2, 3)) / want.abs().max()
print("per-output-channel err (top)", torch.topk(err, 6))
print("z batch var of those channels", var[torch.topk(err, 6).indices], "min var", float(var.min()), "median", float(var.median()))
erri = (got - want).abs().amax(dim=(0, 2, 3)) / want.abs().max()
print("per-INPUT-channel err (top)", torch.topk(erri, 6))
print("x0 var of those input channels", xvar[torch.topk(erri, 6).indices], "min x var", float(xvar.min()))
kb = p + ".layer1.0.bn1.bias"
eb = (tape.param_grads[kb].cpu() - leaves[kb].grad).abs() / leaves[kb].grad.abs().max()
print("bn1.bias err top", torch.topk(eb, 6))
