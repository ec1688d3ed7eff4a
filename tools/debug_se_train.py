"""Debug (GPU): SEBlock / flatten tail in train mode vs oracle autograd, per parameter."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import model_ref as M, train_ref as TR
from thinktwice_amd import autodiff, config, layers, params
from thinktwice_amd.fusion import SEBlock, FlattenTail

cfg = config.model_config()
sd = params.init_params(cfg, seed=4, parts=("fusion",))
g = torch.Generator().manual_seed(1)
for name, C, hw, B in (("MLP10", 64, 10, 4), ("MLP21", 32, 21, 4), ("MLP2", 256, 2, 8)):
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(name + ".") and v.is_floating_point()
              and v.dim() > 0 and not k.endswith(("running_mean", "running_var"))}
    sdr = dict(sd); sdr.update(leaves)
    x = torch.randn(B, C, hw, hw, generator=g).relu()
    xt = x.clone().requires_grad_(True)
    R = torch.randn(B, C, hw, hw, generator=g)
    with TR.train_mode():
        y = M.se_basic_block(sdr, name, xt)
        (y * R).sum().backward()
    blk = SEBlock(sd, name, "cuda")
    xq = x.permute(0, 2, 3, 1).contiguous().cuda()
    layers.BN_TRAIN = True
    with autodiff.Tape(x3=False) as tape:
        out = blk(xq)
        tape.seed(out, R.permute(0, 2, 3, 1))
        tape.backward()
        gx = tape.grad(xq).cpu()
    layers.BN_TRAIN = False
    torch.cuda.synchronize()
    print(name, "fwd", float((out.cpu().permute(0, 3, 1, 2) - y.detach()).abs().max() / y.detach().abs().max()),
          "dx", float((gx.permute(0, 3, 1, 2) - xt.grad).norm() / xt.grad.norm()))
    for k, v in leaves.items():
        got = tape.param_grads[k].cpu()
        print(f"   {k:24s} {float((got - v.grad).norm() / v.grad.norm().clamp_min(1e-20)):.3e}  |g| {float(v.grad.norm()):.3e}")

print("---- flatten tail")
names = [k for k in sd if k.split(".")[0] in ("MLP10", "MLP4", "MLP2", "conv21_10", "conv10_4", "conv4_2", "output_fc")]
leaves = {k: sd[k].clone().requires_grad_(True) for k in names if sd[k].is_floating_point() and sd[k].dim() > 0
          and not k.endswith(("running_mean", "running_var"))}
sdr = dict(sd); sdr.update(leaves)
B = 4
x = torch.randn(B, 32, 21, 21, generator=g).relu()
xt = x.clone().requires_grad_(True)
with TR.train_mode():
    flat, mids = M.flatten_tail(sdr, xt)
    outs = [flat] + mids
    R = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * r).sum() for o, r in zip(outs, R)).backward()
tail = FlattenTail(sd, "cuda")
xq = x.permute(0, 2, 3, 1).contiguous().cuda()
layers.BN_TRAIN = True
with autodiff.Tape(x3=False) as tape:
    hflat, hmids = tail(xq, want_mids=True)
    tape.seed(hflat, R[0])
    for t, r in zip(hmids, R[1:]):
        tape.seed(t, r.permute(0, 2, 3, 1))
    tape.backward()
    gx = tape.grad(xq).cpu()
layers.BN_TRAIN = False
torch.cuda.synchronize()
print("flat fwd", float((hflat.cpu() - flat.detach()).abs().max() / flat.detach().abs().max()),
      "dx", float((gx.permute(0, 3, 1, 2) - xt.grad).norm() / xt.grad.norm()))
for k, v in leaves.items():
    got = tape.param_grads[k].cpu()
    e = float((got - v.grad).norm() / v.grad.norm().clamp_min(1e-20))
    if e > 1e-5 and float(v.grad.norm()) > 1e-3:
        print(f"   {k:24s} {e:.3e}  |g| {float(v.grad.norm()):.3e}")
