"""Debug (GPU): a ResNet bottleneck with train-mode BN through the tape vs torch autograd, progressively."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from thinktwice_amd import autodiff, layers, ops, weights

g = torch.Generator().manual_seed(0)
N, H, W, Cin, Cm, Co = 4, 16, 32, 64, 64, 256
img = torch.randn(N, 3, 4 * H, 4 * W, generator=g)
def mk(co, ci, k):
    return torch.randn(co, ci, k, k, generator=g) * (ci * k * k) ** -0.5
sd = {}
for name, (co, ci, k) in {"stem": (Cin, 3, 7), "c1": (Cm, Cin, 1), "c2": (Cm, Cm, 3), "c3": (Co, Cm, 1), "ds": (Co, Cin, 1)}.items():
    sd[name + ".weight"] = mk(co, ci, k)
    sd[name + "_bn.weight"] = torch.rand(co, generator=g) + 0.5
    sd[name + "_bn.bias"] = torch.randn(co, generator=g) * 0.2
    sd[name + "_bn.running_mean"] = torch.zeros(co)
    sd[name + "_bn.running_var"] = torch.ones(co)
leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if "running" not in k}
def bn(t, n):
    return F.batch_norm(t, None, None, leaves[n + "_bn.weight"], leaves[n + "_bn.bias"], True, 0.0, 1e-5)
y0 = F.relu(bn(F.conv2d(img, leaves["stem.weight"], None, 2, 3), "stem"))
xt = F.max_pool2d(y0, 3, 2, 1)
xt.retain_grad()
y1 = F.relu(bn(F.conv2d(xt, leaves["c1.weight"]), "c1"))
y2 = F.relu(bn(F.conv2d(y1, leaves["c2.weight"], padding=1), "c2"))
idt = bn(F.conv2d(xt, leaves["ds.weight"]), "ds")
y3 = F.relu(bn(F.conv2d(y2, leaves["c3.weight"]), "c3") + idt)
R = torch.randn(y3.shape, generator=g)
(y3 * R).sum().backward()

autodiff.clear_metas()
dev = torch.device("cuda")
L = {n: layers.conv_from_sd(sd, n, torch.float32, dev, bn=n + "_bn", pad=1 if n == "c2" else 0, act="none" if n == "ds" else "relu")
     for n in ("c1", "c2", "c3", "ds")}
L["stem"] = layers.conv_from_sd(sd, "stem", torch.float32, dev, bn="stem_bn", stride=2, pad=3, act="relu", cin_pad=4)
iq = weights.to_channel_last(img, torch.float32).cuda()
layers.BN_TRAIN = True
with autodiff.Tape(x3=False) as tape:
    xq = ops.maxpool3x3s2(L["stem"](iq, stop_grad=True))
    i_ = L["ds"](xq)
    a = L["c2"](L["c1"](xq))
    out = L["c3"](a, res1=i_)
    tape.seed(out, R.permute(0, 2, 3, 1))
    tape.backward()
    gx = tape.grad(xq).cpu()
layers.BN_TRAIN = False
torch.cuda.synchronize()
print("fwd", float((out.cpu().permute(0, 3, 1, 2) - y3.detach()).abs().max() / y3.detach().abs().max()))
print("dx", float((gx.permute(0, 3, 1, 2) - xt.grad).abs().max() / xt.grad.abs().max()))
for k, v in leaves.items():
    got = tape.param_grads[k.replace("_bn", "_bn")].cpu()
    print(f"{k:16s} {float((got - v.grad).abs().max() / v.grad.abs().max()):.3e}")
