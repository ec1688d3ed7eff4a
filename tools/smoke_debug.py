#!/usr/bin/env python
"""smoke_forward.run() with the intermediates compared against the oracle (debug aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def main():
    from oracle import model_ref as M
    from thinktwice_amd import model as tm, params, synth
    hw = (128, 256)
    dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
    m, cfg = tm.build_thinktwice(final_dim=hw, dtype={"f32": torch.float32}.get(dt, dt))
    sd = params.init_params(cfg, seed=0)
    m.load_state_dict(sd)
    batch = synth.make_batch(int(sys.argv[2]) if len(sys.argv) > 2 else 1, img_hw=hw, num_points=8192)
    out = m.forward_inference(tm.batch_to_device(batch))
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = M.forward_inference(sd, cfg, batch, return_intermediates=True)

    def rel(a, b):
        return float((a.float().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-9))
    print("cam_bev", rel(out["_cam_bev_cl"].permute(0, 3, 1, 2), ref["_cam_bev"]))
    print("lidar_bev", rel(out["_lidar_bev_cl"].permute(0, 3, 1, 2), ref["_lidar_bev"]))
    print("seg", rel(out["_seg_cl"][..., :12].permute(0, 3, 1, 2), ref["_cam"]["seg"]))
    print("flat", rel(out["_flat"], ref["_flat"]), "meas", rel(out["_meas"], ref["_meas"]))
    for k in ("pred_wp", "mu_branches", "sigma_branches", "refine_BEV_feature", "pred_speed", "bev_feature", "future_mu"):
        print(k, rel(out[k], ref[k]))


if __name__ == "__main__":
    main()
