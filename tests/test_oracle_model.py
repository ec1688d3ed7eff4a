"""CPU: the oracle restatement vs golden vectors produced by the reference's own modules
(tests/golden/gen_golden.py, run where /root/reference exists), and the state_dict layout."""
import json
import os

import numpy as np
import torch

from oracle import model_ref as M
from thinktwice_amd import config, params, synth


def _check_pack(pack, tensors, prefix="", tol=1e-5):
    n = 0
    for k, v in tensors.items():
        if not torch.is_tensor(v):
            continue
        v = v.detach().float()
        key = prefix + k
        if key in pack.files:
            want = pack[key]
            np.testing.assert_allclose(v.numpy(), want, rtol=tol, atol=tol * max(1.0, float(np.abs(want).max())))
            n += 1
        elif key + "__idx" in pack.files:
            got = v.reshape(-1)[torch.from_numpy(pack[key + "__idx"])].numpy()
            want = pack[key + "__val"]
            np.testing.assert_allclose(got, want, rtol=tol, atol=tol * max(1.0, float(pack[key + "__stats"][2])))
            n += 1
    return n


def test_param_spec_matches_reference_state_dict(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "reference_state_dict_keys.json")))
    cfg = config.model_config(final_dim=(128, 256))
    spec = params.param_spec(cfg)
    third_party = ("img_encoder.img_backbone.", "img_encoder.img_neck.", "lidar_encoder.")
    mine = {k: list(s) for k, (s, _) in spec.items() if not k.startswith(third_party)}
    assert mine == ref
    # third-party modules: torchvision ResNet-50 has 53 convs / 53 BNs; PAFPN 14 convs
    rn = [k for k in spec if k.startswith("img_encoder.img_backbone.") and k.endswith("weight") and len(spec[k][0]) == 4]
    assert len(rn) == 53
    assert sum(int(np.prod(s)) for k, (s, kd) in spec.items() if k.startswith("img_encoder.img_backbone.")
               and kd in ("w", "bn_w", "bn_b")) == 25_557_032 - 2_049_000   # resnet50 minus fc


def test_init_params_is_deterministic_per_key():
    cfg = config.model_config(final_dim=(128, 256))
    a = params.init_params(cfg, seed=0, parts=("fusion",))
    b = params.init_params(cfg, seed=0, parts=("fusion", "decoder"))
    for k in a:
        assert torch.equal(a[k], b[k])
    c = params.init_params(cfg, seed=1, parts=("fusion",))
    assert not torch.equal(a["conv_cam.0.weight"], c["conv_cam.0.weight"])


def test_oracle_forward_matches_reference_golden_small(golden_dir):
    pack = np.load(os.path.join(golden_dir, "f7_forward_small_b2.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    cfg = config.model_config(final_dim=(H, W))
    sd = params.init_params(cfg, seed=seed)
    batch = synth.make_batch(B, img_hw=(H, W), num_points=npts)
    with torch.no_grad():
        out = M.forward_inference(sd, cfg, batch, return_intermediates=True)
    assert _check_pack(pack, out) >= 14
    inter = {"cam_bev": out["_cam_bev"], "lidar_bev": out["_lidar_bev"], "flat": out["_flat"],
             "seg": out["_cam"]["seg"], "depth": out["_cam"]["depth"], "context": out["_cam"]["context"],
             "fpn0": out["_cam"]["fpn_feats"][0], "fpn3": out["_cam"]["fpn_feats"][3]}
    assert _check_pack(pack, inter, prefix="inter__") == 8
    np.testing.assert_array_equal(np.array([i["max_len"] for i in out["_look_info"]]), pack["look_max_len"])
    assert float(pack["oracle_vs_reference_worst_rel_err"][0]) < 1e-4
    # output contract
    assert out["pred_wp"].shape == (B, 6, 4, 2) and out["mu_branches"].shape == (B, 6, 2)
    assert out["refine_future_BEV_feature"].shape == (B, 5, 4, 32, 21, 21)


def test_oracle_train_losses_match_reference_golden(golden_dir):
    """SURVEY 8f-4 groundwork: the training forward (teacher-forcing pass + all 23 loss terms + seg focal loss + depth
    BCE) of oracle/train_ref.py against the reference's own `forward_train` (model.eval(), F10)."""
    from oracle import train_ref as TR
    pack = np.load(os.path.join(golden_dir, "f10_train_losses_b2.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    cfg = config.model_config(final_dim=(H, W))
    sd = params.init_params(cfg, seed=seed)
    batch = synth.make_batch(B, img_hw=(H, W), num_points=npts)
    batch.update(synth.make_train_targets(B, img_hw=(H, W)))
    with torch.no_grad():
        losses, pred = TR.forward_train(sd, cfg, batch)
    names = [k for k in pack.files if k not in ("meta", "oracle_vs_reference_worst_rel_err")]
    assert len(names) == 23 and set(names) == set(losses)
    for k in names:
        got, want = losses[k].detach().float().numpy(), pack[k]
        assert got.shape == want.shape, k                      # value_loss keeps the reference's (B,1) shape
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7, err_msg=k)
    assert float(pack["oracle_vs_reference_worst_rel_err"][0]) == 0.0
    # teacher-forcing outputs (DEC:525-532)
    assert pred["teacher_pred_wp_offset"].shape == (B, 5, 4, 2)
    assert pred["teacher_pred_ctrl_offset_lis"].shape == (B, 5, 4, 4)
    assert pred["teacher_future_BEV_feature"].shape == (B, 5, 4, 32, 21, 21)
    # the supervised parts respond to the supervision: perfect expert waypoints zero the wp-offset statistic
    b2 = dict(batch)
    b2["waypoints"] = pred["pred_wp"][:, -1].clone()
    with torch.no_grad():
        l2 = TR.decoder_loss(cfg, b2, pred, [None, None, pred["bev_feature"], *[torch.zeros(1)] * 3][:3] +
                             [batch["grid_feature"][i] for i in (3, 4, 5)])
    assert float(l2["longitudinal_offset"]) == 0.0 and float(l2["lateral_offset"]) == 0.0


def test_oracle_train_mode_losses_match_reference_golden(golden_dir):
    """As above under model.train(): batch-statistics BatchNorm (per sweep) and the live ASPP Dropout(0.5), same torch
    seed as the generator (F11)."""
    from oracle import train_ref as TR
    pack = np.load(os.path.join(golden_dir, "f11_train_losses_trainmode_b2.npz"))
    B, H, W, npts, seed, rng, jitter = (int(v) for v in pack["meta"])
    cfg = config.model_config(final_dim=(H, W))
    sd = params.init_params(cfg, seed=seed)
    batch = synth.make_batch(B, img_hw=(H, W), num_points=npts, jitter_calib=jitter)   # (see synth.make_img_metas)
    batch.update(synth.make_train_targets(B, img_hw=(H, W)))
    with torch.no_grad(), TR.train_mode():
        torch.manual_seed(rng)
        losses, _ = TR.forward_train(sd, cfg, batch)
    assert M.TRAIN_MODE is False                                   # the switch is scoped
    names = [k for k in pack.files if k not in ("meta", "oracle_vs_reference_worst_rel_err")]
    assert len(names) == 23
    for k in names:
        np.testing.assert_allclose(losses[k].detach().float().numpy(), pack[k], rtol=1e-5, atol=1e-7, err_msg=k)
    # and it is a different computation from the running-statistics one
    eval_pack = np.load(os.path.join(golden_dir, "f10_train_losses_b2.npz"))
    assert abs(float(pack["speed_loss"]) - float(eval_pack["speed_loss"])) > 0.1


def test_oracle_backward_matches_reference_gradients(golden_dir):
    """F13: autograd through the oracle restatement == the reference's own backward (custom VoxelPooling Function,
    detach / no_grad placements, `_parse_losses`): gradient norm + 8 sampled entries of all 878 live parameters, and
    the same 90 dead parameters.  This is the gradient oracle the backward kernels of SURVEY 8f-4 will be checked
    against."""
    from oracle import train_ref as TR
    pack = np.load(os.path.join(golden_dir, "f13_train_gradients_b2.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    cfg = config.model_config(final_dim=(H, W))
    sd = params.init_params(cfg, seed=seed)
    batch = synth.make_batch(B, img_hw=(H, W), num_points=npts)
    batch.update(synth.make_train_targets(B, img_hw=(H, W)))
    skip = ("running_mean", "running_var", "num_batches_tracked", "voxel_size", "voxel_coord", "voxel_num", "frustum")
    leaves = {k: (v.clone().requires_grad_() if v.is_floating_point() and not k.endswith(skip) else v)
              for k, v in sd.items()}
    losses, _ = TR.forward_train(leaves, cfg, batch)
    loss = TR.total_loss(losses)
    assert abs(float(loss.detach()) - float(pack["total_loss"][0])) < 1e-3 * abs(float(pack["total_loss"][0]))
    loss.backward()
    dead = sorted(k for k, v in leaves.items() if torch.is_tensor(v) and v.requires_grad and v.grad is None)
    assert dead == sorted(str(k) for k in pack["dead"]) and len(dead) == 90
    assert all(".prediction_module.ffn." in k or "lidar_look_module" in k or "look_feature_MLP" in k for k in dead)
    worst = 0.0
    for name, norm, idx, smp in zip(pack["names"], pack["norms"], pack["idx"], pack["samples"]):
        g = leaves[str(name)].grad
        assert g is not None, name
        worst = max(worst, abs(float(g.norm()) - float(norm)) / max(float(norm), 1e-12))
        got = g.reshape(-1)[torch.from_numpy(idx)].numpy()
        assert np.allclose(got, smp, rtol=2e-3, atol=1e-4 * float(norm) + 1e-9), (name, got, smp)
    assert worst < 1e-3, worst
    assert float(pack["oracle_vs_reference_worst_rel_err"][0]) < 1e-4


def test_sca_batch_coupling_quirk():
    """MSDA:338-341: outputs depend on the local batch size (first B slots zeroed, / B)."""
    cfg = config.model_config(final_dim=(128, 256), refine_num=1)
    sd = params.init_params(cfg, seed=0, parts=("fusion", "decoder"))
    g = torch.Generator().manual_seed(0)
    B = 3
    flat, bev, meas = torch.randn(B, 256, generator=g), torch.randn(B, 32, 21, 21, generator=g), torch.randn(B, 128, generator=g)
    fpn = [torch.randn(B * 4, 256, 32 >> i, 64 >> i, generator=g) for i in range(4)]
    metas = synth.make_img_metas(B, final_dim=(128, 256))
    from oracle import lss_geometry as og
    _, _, _, l2i, ida = og.assemble_camera_mats(metas)
    with torch.no_grad():
        full = M.decoder_forward(sd, cfg, flat, bev, meas, l2i, ida, fpn)
        one = M.decoder_forward(sd, cfg, flat[:1], bev[:1], meas[:1], l2i[:1], ida[:1], [f[:4] for f in fpn])
    # coarse heads are per-sample ...
    assert torch.allclose(full["pred_wp"][:1, 0], one["pred_wp"][:, 0], atol=1e-5)
    # ... the refined stage is batch-coupled through the SCA normalisation
    assert not torch.allclose(full["pred_wp"][:1, 1], one["pred_wp"][:, 1], atol=1e-5)
