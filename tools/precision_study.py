"""CPU study of reduced-precision STORAGE in the camera trunk (tools only; uses the oracle as the f32 truth).

Emulates what the HIP trunk does in a 16-bit mode -- weights and every stored activation rounded to the storage type,
products accumulated in f32 -- by patching the oracle's conv primitives, and prints the rel-max error of the
LSS outputs per policy.  Used to pick the precision mix whose outputs meet the 1e-3 tolerance;
the GPU tests are the proof, this is the map.

    python tools/precision_study.py [policy ...]      policies: bf16 f16 f16w2 f16a2 split ...
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from oracle import model_ref as M  # noqa: E402


def q(t, dt):
    return t if dt is None else t.to(dt).float()


def split2(t, dt):
    """hi + lo pair in `dt` (what a 2-term split operand carries)."""
    hi = t.to(dt).float()
    lo = (t - hi).to(dt).float()
    return hi + lo


class Policy:
    """act / wgt: storage dtype of activations / weights (None = f32); *_split: keep hi+lo pairs."""

    def __init__(self, act=None, wgt=None, act_split=False, wgt_split=False, f32_layers=(), w2_layers=()):
        self.act, self.wgt, self.act_split, self.wgt_split = act, wgt, act_split, wgt_split
        self.f32_layers = tuple(f32_layers)
        self.w2_layers = tuple(w2_layers)

    def qa(self, x):
        if self.act is None:
            return x
        return split2(x, self.act) if self.act_split else q(x, self.act)

    def qw(self, w):
        if self.wgt is None:
            return w
        return split2(w, self.wgt) if self.wgt_split else q(w, self.wgt)


POLICIES = {
    "f32": Policy(),
    "bf16": Policy(torch.bfloat16, torch.bfloat16),
    "f16": Policy(torch.float16, torch.float16),
    "f16_w2": Policy(torch.float16, torch.float16, wgt_split=True),          # weights hi+lo (2 MFMAs)
    "f16_a2": Policy(torch.float16, torch.float16, act_split=True),          # activations hi+lo
    "f16_split": Policy(torch.float16, torch.float16, act_split=True, wgt_split=True),
    "bf16_split": Policy(torch.bfloat16, torch.bfloat16, act_split=True, wgt_split=True),
    "f16_act_only": Policy(torch.float16, None),
    "f16_wgt_only": Policy(None, torch.float16),
}


def run(policy, sd, cfg, batch, prefix_f32=()):
    orig_conv, orig_ct, orig_mm = M.conv, F.conv_transpose2d, torch.matmul
    pol = policy

    def conv(sd_, p, x, stride=1, padding=0, dilation=1, groups=1):
        if any(p.startswith(s) or s in p for s in pol.f32_layers):
            return orig_conv(sd_, p, x, stride, padding, dilation, groups)
        w = sd_[p + ".weight"]
        w = split2(w, pol.wgt) if any(s in p for s in pol.w2_layers) else pol.qw(w)
        return F.conv2d(pol.qa(x), w, sd_.get(p + ".bias"), stride, padding, dilation, groups)

    def conv_t(x, w, b=None, stride=1, *a, **k):
        return orig_ct(pol.qa(x), pol.qw(w), b, stride, *a, **k)

    M.conv = conv
    F.conv_transpose2d = conv_t
    try:
        with torch.no_grad():
            return M.lss_forward(sd, "img_encoder", cfg, batch["img"], batch["img_metas"])
    finally:
        M.conv, F.conv_transpose2d = orig_conv, orig_ct


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def rms(a, b):
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12))


def main():
    from thinktwice_amd import config, params, synth
    hw = (128, 256)
    cfg = config.model_config(final_dim=hw)
    sd = params.init_params(cfg, seed=0, parts=("img_encoder",))
    batch = synth.make_batch(2, img_hw=hw, num_points=1000)
    names = sys.argv[1:] or ["bf16", "f16", "f16_w2", "f16_a2", "f16_split"]
    ref = run(POLICIES["f32"], sd, cfg, batch)
    for n in names:
        if "@w2:" in n:   # policy@w2:layer-substring,...: those layers keep hi+lo weights
            base, layers = n.split("@w2:")
            b = POLICIES[base]
            pol = Policy(b.act, b.wgt, b.act_split, b.wgt_split, (), layers.split(","))
        elif "@" in n:    # policy@layer-substring,layer-substring: those layers stay f32
            base, layers = n.split("@")
            b = POLICIES[base]
            pol = Policy(b.act, b.wgt, b.act_split, b.wgt_split, layers.split(","))
        else:
            pol = POLICIES[n]
        out = run(pol, sd, cfg, batch)
        errs = {f"fpn{i}": rel(out["fpn_feats"][i], ref["fpn_feats"][i]) for i in range(4)}
        for k in ("seg", "depth", "bev"):
            errs[k] = rel(out[k], ref[k])
        errs["bev_rms"] = rms(out["bev"], ref["bev"])
        print(f"{n:14s}", " ".join(f"{k}={v:.2e}" for k, v in errs.items()), flush=True)


if __name__ == "__main__":
    main()
