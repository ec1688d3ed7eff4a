"""CPU study for a per-stage precision plan (tools only, the oracle is the f32 truth).

Question: can some stages of the forward run a TWO-MFMA product -- activations rounded to IEEE half at the operand
(one `v_cvt` per element pair, no hi / lo split), weights kept as an f16 (hi, lo) pair, f32 accumulation and f32
storage -- instead of the bf16x3 product (three MFMAs, 2^-17 per product), and keep all 14 outputs within the
1e-3 bound with margin?  The emulation patches the oracle's conv / linear primitives per stage:

    a16   : x -> f16 -> f32 at the operand, weights hi + lo in f16 (22 bits, ~ exact), f32 sums      [2 MFMAs]
    a16w16: x and w both rounded to f16 once                                                          [1 MFMA]
    abf   : x -> bf16 at the operand, weights exact                                                   (for scale)

Every other stage stays exact f32 (bf16x3 is 2^-17 per product: below everything measured here).

    python tools/precision_plan.py [--hw 448 896] [--batch 1] [--points 65536] policy[@stage,stage...] ...

Stages (substring of the parameter prefix): layer1 layer2 layer3 layer4 conv1(stem) img_neck neck_conv depth_net seg_net
seg_res merge_seg lidar fusion decoder ...; `a16@layer1,layer2` applies a16 to those stages only, `a16@-layer4` to
everything in the camera encoder except layer4.  Prints the rel-max error of the 14 output keys, seg and camera BEV.
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from oracle import model_ref as M  # noqa: E402

KEYS = ("pred_wp", "mu_branches", "sigma_branches", "future_mu", "future_sigma", "pred_speed", "pred_value_traj",
        "pred_value_ctrl", "pred_features_traj", "pred_features_ctrl", "bev_feature", "refine_BEV_feature",
        "refine_flattned_BEV_feature", "refine_future_BEV_feature")


def split2(t, dt):
    hi = t.to(dt).float()
    return hi + (t - hi).to(dt).float()


def make_q(mode):
    if mode == "a16":
        return (lambda x: x.half().float()), (lambda w: split2(w, torch.float16))
    if mode == "a16w16":
        return (lambda x: x.half().float()), (lambda w: w.half().float())
    if mode == "abf":
        return (lambda x: x.bfloat16().float()), (lambda w: w)
    if mode == "abf2":      # bf16 hi + lo activations x bf16 hi weights  (2 MFMAs the other way round)
        return (lambda x: split2(x, torch.bfloat16)), (lambda w: w.bfloat16().float())
    raise SystemExit(f"unknown mode {mode}")


def selected(p, stages):
    """stages: list of substrings; a leading '-' excludes. Empty list = every camera-encoder conv."""
    inc = [s for s in stages if not s.startswith("-")]
    exc = [s[1:] for s in stages if s.startswith("-")]
    if any(s in p for s in exc):
        return False
    if inc:
        return any(s in p for s in inc)
    return p.startswith("img_encoder")


def run(sd, cfg, batch, mode=None, stages=()):
    orig_conv, orig_linear = M.conv, M.linear
    hits = set()
    if mode is not None:
        qa, qw = make_q(mode)

        def conv(sd_, p, x, stride=1, padding=0, dilation=1, groups=1):
            if not selected(p, stages):
                return orig_conv(sd_, p, x, stride, padding, dilation, groups)
            hits.add(p)
            return F.conv2d(qa(x), qw(sd_[p + ".weight"]), sd_.get(p + ".bias"), stride, padding, dilation, groups)

        def linear(sd_, p, x):
            if not selected(p, stages):
                return orig_linear(sd_, p, x)
            hits.add(p)
            return F.linear(qa(x), qw(sd_[p + ".weight"]), sd_.get(p + ".bias"))

        M.conv, M.linear = conv, linear
    try:
        with torch.no_grad():
            out = M.forward_inference(sd, cfg, batch, return_intermediates=True)
    finally:
        M.conv, M.linear = orig_conv, orig_linear
    return out, hits


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, nargs=2, default=(448, 896))
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("policies", nargs="*")
    a = ap.parse_args()
    from thinktwice_amd import config, params, synth
    cfg = config.model_config(final_dim=tuple(a.hw))
    sd = params.init_params(cfg, seed=a.seed)
    batch = synth.make_batch(a.batch, img_hw=tuple(a.hw), num_points=a.points)
    t0 = time.time()
    ref, _ = run(sd, cfg, batch)
    print(f"# f32 truth: B={a.batch} {a.hw[0]}x{a.hw[1]} {a.points} points, {time.time() - t0:.0f} s per forward", flush=True)
    for pol in a.policies or ["a16"]:
        mode, _, st = pol.partition("@")
        stages = [s for s in st.split(",") if s]
        out, hits = run(sd, cfg, batch, mode, stages)
        errs = {k: rel(out[k], ref[k]) for k in KEYS}
        worst = max(errs, key=errs.get)
        cam = {"seg": rel(out["_cam"]["seg"], ref["_cam"]["seg"]), "cam_bev": rel(out["_cam_bev"], ref["_cam_bev"]),
               "fpn": max(rel(x, y) for x, y in zip(out["_cam"]["fpn_feats"], ref["_cam"]["fpn_feats"]))}
        print(f"{pol:34s} layers={len(hits):3d} pred_wp={errs['pred_wp']:.2e} worst={errs[worst]:.2e} ({worst}) "
              + " ".join(f"{k}={v:.2e}" for k, v in cam.items()), flush=True)
        print("    " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)


if __name__ == "__main__":
    main()
