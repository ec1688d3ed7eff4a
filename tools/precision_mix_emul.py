"""CPU emulation of the MIXED precision / storage plan at STORAGE level, scored against the golden packs.

The plan ("f32x3h" mode of the product): ResNet layer1 + layer2 and the PAFPN keep their activations in IEEE half in HBM and
multiply them with an f16 (hi, lo) weight pair -- two MFMAs per product, exact with respect to the STORED operands
(11 + 22 bits < 2^-20 per product) -- everything else stays bf16x3 on f32 storage (2^-17 per product, exact here).  What costs
accuracy is therefore the rounding of the STORED tensors, which tools/precision_plan.py did not emulate (it rounded at the
operand and kept the residual identity in f32).  Here every tensor the plan stores in half is rounded where the kernel would
store it:

  maxpool output                      -> f16   (layer1's input and first identity)
  layer1.*, layer2.*  conv outputs    -> f16   (after the folded BN / residual / ReLU of the epilogue; identity read back as f16)
  layer3.0.conv1 / .downsample        read C3 as stored (f16): exact product, f32 output
  PAFPN lateral outputs               -> f16   (lateral 2 / 3 read the f32 C4 / C5 through bf16x3 and store f16)
  top-down nearest-upsample adds      -> f16   (f32 add of two f16 values, rounded once)
  fpn_convs.i outputs (i >= 1)        -> f16   ; fpn_convs.0: f32 to the UNet concat buffer AND an f16 copy for downsample_convs.0
  bottom-up adds (downsample epilogue)-> f16
  pafpn_convs outputs                 -> f32   (the PAFPN's outputs: consumers are bf16x3 / the decoder's value projection)

Variants (`--plan`): `l1l2neck` (above), `l1neck` (layer2 stays f32 / bf16x3), `l1l2` (PAFPN stays f32), `neck`.

    python tools/precision_mix_emul.py [--plan l1l2neck] [--packs f7 f8 f14]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from oracle import model_ref as M  # noqa: E402

KEYS = ("pred_wp", "mu_branches", "sigma_branches", "future_mu", "future_sigma", "pred_speed", "pred_value_traj",
        "pred_value_ctrl", "pred_features_traj", "pred_features_ctrl", "bev_feature", "refine_BEV_feature",
        "refine_flattned_BEV_feature", "refine_future_BEV_feature")
PACKS = {"f7": "f7_forward_small_b2.npz", "f8": "f8_forward_full_b1.npz", "f14": "f14_forward_full_b8.npz"}


def q(x):
    return x.half().float()


def make_patches(plan):
    h_l1 = "l1" in plan
    h_l2 = "l2" in plan
    h_neck = "neck" in plan
    conv, bn = M.conv, M.bn

    def resnet50(sd, p, x):
        x = F.relu(bn(sd, p + ".bn1", conv(sd, p + ".conv1", x, 2, 3)))
        x = F.max_pool2d(x, 3, 2, 1)
        outs = []
        for li, blocks in enumerate((3, 4, 6, 3), start=1):
            half = (li == 1 and h_l1) or (li == 2 and h_l2)
            st = q if half else (lambda t: t)
            if half:
                x = st(x)           # the stage's input is stored in half (maxpool output / previous half stage: already rounded)
            for b in range(blocks):
                qn = f"{p}.layer{li}.{b}"
                stride = 2 if (b == 0 and li > 1) else 1
                idt = x
                y = st(F.relu(bn(sd, qn + ".bn1", conv(sd, qn + ".conv1", x))))
                y = st(F.relu(bn(sd, qn + ".bn2", conv(sd, qn + ".conv2", y, stride, 1))))
                y = bn(sd, qn + ".bn3", conv(sd, qn + ".conv3", y))
                if b == 0:
                    idt = st(bn(sd, qn + ".downsample.1", conv(sd, qn + ".downsample.0", x, stride)))
                x = st(F.relu(y + idt))
            outs.append(x)
        return outs

    def pafpn_operand_only(sd, p, feats):
        """plan `neckop`: the PAFPN's sums stay in f32 (laterals, top-down and bottom-up chains); only the INPUT of each 3 x 3
        conv is a half copy (one rounding per conv input, none accumulating along the chains)."""
        n = len(feats)
        lat = [conv(sd, f"{p}.lateral_convs.{i}.conv", feats[i]) for i in range(n)]
        for i in range(n - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        inter = [conv(sd, f"{p}.fpn_convs.{i}.conv", q(lat[i]), 1, 1) for i in range(n)]
        for i in range(n - 1):
            inter[i + 1] = inter[i + 1] + conv(sd, f"{p}.downsample_convs.{i}.conv", q(inter[i]), 2, 1)
        outs = [inter[0]]
        for i in range(1, n):
            outs.append(conv(sd, f"{p}.pafpn_convs.{i - 1}.conv", q(inter[i]), 1, 1))
        return outs

    def pafpn(sd, p, feats):
        if "neckop" in plan:
            return pafpn_operand_only(sd, p, feats)
        if not h_neck:
            return M_ORIG["pafpn"](sd, p, feats)
        n = len(feats)
        lat = [q(conv(sd, f"{p}.lateral_convs.{i}.conv", feats[i])) for i in range(n)]
        for i in range(n - 1, 0, -1):
            lat[i - 1] = q(lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest"))
        inter32 = [conv(sd, f"{p}.fpn_convs.{i}.conv", lat[i], 1, 1) for i in range(n)]
        out0 = inter32[0]                                 # f32 copy: the PAFPN's first output
        inter = [q(t) for t in inter32]
        for i in range(n - 1):
            inter[i + 1] = q(inter[i + 1] + conv(sd, f"{p}.downsample_convs.{i}.conv", inter[i], 2, 1))
        outs = [out0]
        for i in range(1, n):
            outs.append(conv(sd, f"{p}.pafpn_convs.{i - 1}.conv", inter[i], 1, 1))
        return outs

    return resnet50, pafpn


M_ORIG = {"resnet50": M.resnet50, "pafpn": M.pafpn}


def score(pack, out):
    errs = {}
    for k in KEYS:
        v = out[k].detach().float().cpu()
        if k in pack.files:
            want = torch.from_numpy(pack[k])
            errs[k] = float((v - want).abs().max() / want.abs().max().clamp_min(1e-6))
        else:
            idx = torch.from_numpy(pack[k + "__idx"])
            want = torch.from_numpy(pack[k + "__val"])
            errs[k] = float((v.reshape(-1)[idx] - want).abs().max() / float(pack[k + "__stats"][2]))
    inter = {}
    if "inter__seg__idx" in pack.files:
        seg = out["_cam"]["seg"].float().reshape(-1)
        bev = out["_cam_bev"].float().reshape(-1)
        for name, v in (("seg", seg), ("cam_bev", bev)):
            idx = torch.from_numpy(pack[f"inter__{name}__idx"])
            want = torch.from_numpy(pack[f"inter__{name}__val"])
            inter[name] = float((v[idx] - want).abs().max() / float(pack[f"inter__{name}__stats"][2]))
    if "pred_wp" in pack.files:
        inter["wp_L2_mm"] = 1e3 * float((out["pred_wp"].float() - torch.from_numpy(pack["pred_wp"])).norm(dim=-1).max())
    flips = None
    return errs, inter, flips


def seeds_mode(a):
    """--seeds s0 s1 ...: B = --batch samples at the full size, weights init_params(seed), the exact oracle as truth (pred_value_traj
    is one near-cancelling scalar per sample: its relative error is a random draw per (weights, batch), so one fixture says little)."""
    from thinktwice_amd import config, params, synth
    H, W = a.hw
    cfg = config.model_config(final_dim=(H, W))
    for seed in a.seeds:
        sd = params.init_params(cfg, seed=seed)
        torch.manual_seed(seed)
        batch = synth.make_batch(a.batch, img_hw=(H, W), num_points=a.points, seed=seed) if "seed" in synth.make_batch.__code__.co_varnames \
            else synth.make_batch(a.batch, img_hw=(H, W), num_points=a.points)
        with torch.no_grad():
            ref = M.forward_inference(sd, cfg, batch, return_intermediates=True)
        for plan in a.plan:
            M.resnet50, M.pafpn = make_patches(plan)
            try:
                with torch.no_grad():
                    out = M.forward_inference(sd, cfg, batch, return_intermediates=True)
            finally:
                M.resnet50, M.pafpn = M_ORIG["resnet50"], M_ORIG["pafpn"]
            errs = {k: float((out[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-6)) for k in KEYS}
            worst = max(errs, key=errs.get)
            others = max(v for k, v in errs.items() if k != "pred_value_traj")
            seg = float((out["_cam"]["seg"] - ref["_cam"]["seg"]).abs().max() / ref["_cam"]["seg"].abs().max())
            print(f"seed={seed} B={a.batch} {H}x{W} plan={plan:10s} worst={errs[worst]:.2e} ({worst}) value_traj={errs['pred_value_traj']:.2e} "
                  f"(|v|max={float(ref['pred_value_traj'].abs().max()):.3f}) other_keys<={others:.2e} seg={seg:.2e}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plan", nargs="*", default=["l1l2neck"])
    ap.add_argument("--packs", nargs="*", default=["f7", "f8", "f14"])
    ap.add_argument("--seeds", type=int, nargs="*", default=None)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--hw", type=int, nargs=2, default=(448, 896))
    ap.add_argument("--points", type=int, default=65536)
    a = ap.parse_args()
    if a.seeds:
        return seeds_mode(a)
    from thinktwice_amd import config, params, synth
    for pk in a.packs:
        pack = np.load(os.path.join(ROOT, "tests", "golden", PACKS[pk]))
        B, H, W, npts, seed = (int(v) for v in pack["meta"])
        cfg = config.model_config(final_dim=(H, W))
        sd = params.init_params(cfg, seed=seed)
        batch = synth.make_batch(B, img_hw=(H, W), num_points=npts)
        for plan in a.plan:
            t0 = time.time()
            if plan != "exact":
                M.resnet50, M.pafpn = make_patches(plan)
            try:
                with torch.no_grad():
                    out = M.forward_inference(sd, cfg, batch, return_intermediates=True)
            finally:
                M.resnet50, M.pafpn = M_ORIG["resnet50"], M_ORIG["pafpn"]
            errs, inter, _ = score(pack, out)
            worst = max(errs, key=errs.get)
            print(f"{pk} B={B} {H}x{W} plan={plan:10s} worst={errs[worst]:.2e} ({worst}) pred_wp={errs['pred_wp']:.2e} "
                  + " ".join(f"{k}={v:.2e}" for k, v in inter.items()) + f"   [{time.time() - t0:.0f} s]", flush=True)
            print("    " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)


if __name__ == "__main__":
    main()
