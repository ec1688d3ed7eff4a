"""Fusion neck + 5-stage look-and-predict decoder: HIP vs oracle on identical
seeded inputs; B in {1,2,3} because the SCA normalisation couples samples in a batch (MSDA:338-341)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def _inputs(B, hw=(128, 256), seed=0):
    from oracle import lss_geometry as og
    from thinktwice_amd import synth
    g = torch.Generator().manual_seed(seed)
    cam_bev = torch.randn(B, 256, 21, 21, generator=g) * 0.3
    lidar = torch.randn(B, 512, 84, 84, generator=g).abs() * 0.5
    fpn = [torch.randn(B * 4, 256, hw[0] // s, hw[1] // s, generator=g) for s in (4, 8, 16, 32)]
    batch = synth.make_batch(B, img_hw=hw, num_points=16, with_img=False)
    _, _, _, l2i, ida = og.assemble_camera_mats(batch["img_metas"])
    return cam_bev, lidar, fpn, batch, l2i, ida


@pytest.mark.parametrize("fused", [False, True], ids=["layerwise", "composite"])
@pytest.mark.parametrize("B", [1, 2, 3, -2])
def test_fusion_and_decoder_match_oracle(B, fused, monkeypatch):
    """B < 0: |B| samples with a degenerate lidar2img (all zeros) -> no look point projects into any camera, every
    (sample, cam) hit count and max_len are 0 and the look feature is the pure-bias path."""
    no_hits = B < 0
    B = abs(B)
    # composite = decoder_fused.py (~16 launches per layer, bf16x3 arithmetic); layerwise = one launch per op, exact f32
    monkeypatch.setenv("TT_DEC_FUSED", "1" if fused else "0")
    from oracle import model_ref as M
    from thinktwice_amd import config, params, weights
    from thinktwice_amd.encoder_decoder import EncoderDecoder
    from thinktwice_amd.fusion import BEVFusion
    from thinktwice_amd.decoder import ThinkTwiceDecoder
    from thinktwice_amd.layers import linear_from_sd
    hw = (128, 256)
    cfg = config.model_config(final_dim=hw)
    sd = params.init_params(cfg, seed=0, parts=("fusion", "decoder"))
    cam_bev, lidar, fpn, batch, l2i, ida = _inputs(B, hw)
    if no_hits:
        l2i = torch.zeros_like(l2i)
    with torch.no_grad():
        meas_r = M.measurement_feat(sd, batch)
        flat_r, bev_r, _ = M.fusion(sd, cam_bev, lidar)
        ref = M.decoder_forward(sd, cfg, flat_r, bev_r, meas_r, l2i, ida, fpn)

    class Parent:   # the decoder only needs parent_module.flatten_tail
        pass
    dev = torch.device("cuda")
    par = EncoderDecoder.__new__(EncoderDecoder)
    par.device = dev
    par.fusion = BEVFusion(sd, dev)
    par.meas0 = linear_from_sd(sd, "measurements_encoder.0", dev, act="relu", in_pad=12)
    par.meas2 = linear_from_sd(sd, "measurements_encoder.2", dev, act="relu")
    meas = par.measurement_feat(batch)
    flat, bev32, _ = par.fusion(weights.to_channel_last(cam_bev).cuda(), weights.to_channel_last(lidar).cuda())
    dec = ThinkTwiceDecoder(config=cfg["cfg"], bev_h=21, bev_w=21).load_state_dict(sd)
    fpn_cl = [(weights.to_channel_last(f).cuda(), 0, 256) for f in fpn]
    out = dec(flat, bev32, meas, batch["target_point"], par, None, [l2i, ida, fpn_cl, None])
    torch.cuda.synchronize()
    assert _rel(meas.cpu(), meas_r) < 1e-4
    assert _rel(flat.cpu(), flat_r) < 1e-4
    # look-module bookkeeping is integer work: bit exact
    assert (dec.fused is not None) == fused
    for L in range(5):
        cnt, ml = out["_look_info"][L]
        np.testing.assert_array_equal(cnt.cpu().numpy(), ref["_look_info"][L]["count"].numpy())
        assert int(ml.item()) == ref["_look_info"][L]["max_len"]
    errs = {}
    for k in ("pred_wp", "mu_branches", "sigma_branches", "future_mu", "future_sigma", "pred_speed",
              "pred_value_traj", "pred_value_ctrl", "pred_features_traj", "pred_features_ctrl", "bev_feature",
              "refine_BEV_feature", "refine_flattned_BEV_feature", "refine_future_BEV_feature"):
        assert out[k].shape == ref[k].shape, k
        errs[k] = _rel(out[k].cpu(), ref[k])
    print(B, errs)
    assert max(errs.values()) < 1e-3, errs


@pytest.mark.parametrize("fused", [False, True], ids=["layerwise", "composite"])
def test_decoder_teacher_forcing_pass_matches_oracle(fused, monkeypatch):
    """SURVEY 8f-4 (forward half): the teacher-forcing pass (DEC:491-533) -- five layers fed the expert waypoints and
    inv_softplus(expert Beta parameters) -- against the oracle, which reproduces the reference's training forward
    bit-exactly (golden F10).  The ordinary outputs must be unaffected by the extra pass."""
    from oracle import model_ref as M
    from thinktwice_amd import config, params, synth, weights
    from thinktwice_amd.encoder_decoder import EncoderDecoder
    from thinktwice_amd.fusion import BEVFusion
    from thinktwice_amd.decoder import ThinkTwiceDecoder
    from thinktwice_amd.layers import linear_from_sd
    monkeypatch.setenv("TT_DEC_FUSED", "1" if fused else "0")
    B, hw = 2, (128, 256)
    cfg = config.model_config(final_dim=hw)
    sd = params.init_params(cfg, seed=0, parts=("fusion", "decoder"))
    cam_bev, lidar, fpn, batch, l2i, ida = _inputs(B, hw)
    tgt = synth.make_train_targets(B, img_hw=hw)
    teacher = {k: tgt[k] for k in ("waypoints", "action_mu", "action_sigma", "future_action_mu", "future_action_sigma")}
    with torch.no_grad():
        meas_r = M.measurement_feat(sd, batch)
        flat_r, bev_r, _ = M.fusion(sd, cam_bev, lidar)
        ref = M.decoder_forward(sd, cfg, flat_r, bev_r, meas_r, l2i, ida, fpn, teacher=teacher)
    dev = torch.device("cuda")
    par = EncoderDecoder.__new__(EncoderDecoder)
    par.device = dev
    par.fusion = BEVFusion(sd, dev)
    par.meas0 = linear_from_sd(sd, "measurements_encoder.0", dev, act="relu", in_pad=12)
    par.meas2 = linear_from_sd(sd, "measurements_encoder.2", dev, act="relu")
    meas = par.measurement_feat(batch)
    flat, bev32, _ = par.fusion(weights.to_channel_last(cam_bev).cuda(), weights.to_channel_last(lidar).cuda())
    dec = ThinkTwiceDecoder(config=cfg["cfg"], bev_h=21, bev_w=21).load_state_dict(sd)
    fpn_cl = [(weights.to_channel_last(f).cuda(), 0, 256) for f in fpn]
    t_dev = {k: ([t.cuda() for t in v] if isinstance(v, list) else v.cuda()) for k, v in teacher.items()}
    out = dec(flat, bev32, meas, batch["target_point"], par, t_dev, [l2i, ida, fpn_cl, None])
    torch.cuda.synchronize()
    errs = {}
    for k in ("teacher_pred_wp_offset", "teacher_pred_ctrl_offset_lis", "teacher_future_BEV_feature",
              "teacher_refine_flattned_BEV_feature", "teacher_refine_BEV_feature", "pred_wp", "refine_BEV_feature"):
        assert out[k].shape == ref[k].shape, (k, out[k].shape, ref[k].shape)
        errs[k] = _rel(out[k].cpu(), ref[k])
    print(errs)
    assert max(errs.values()) < 1e-3, errs


def test_decoder_with_fused_concat_matches_oracle(monkeypatch):
    """A/B path TT_DEC_FUSED_CONCAT=1: every concatenated MLP input assembled by one tt_concat_rows launch."""
    import thinktwice_amd.decoder as D
    from thinktwice_amd import ops
    # the kernel alone against torch.cat on the decoder's widest case (5 pieces, broadcast + modulo mappings)
    B = 3
    g = torch.Generator().manual_seed(1)
    fflat, look, meas, temporal = (torch.randn(*s, generator=g).cuda() for s in ((B * 4, 256), (B, 256), (B, 128), (4, 128)))
    hin = torch.full((B * 4, 1024), float("nan"), device="cuda")
    ops.concat_rows(hin, [(fflat, 256, 1, 0), (look, 256, 4, 0), (None, 256, 1, 0), (temporal, 128, 1, 4), (meas, 128, 4, 0)])
    want = torch.cat([fflat.view(B, 4, 256), look.unsqueeze(1).expand(B, 4, 256), torch.zeros(B, 4, 256, device="cuda"),
                      temporal.unsqueeze(0).expand(B, 4, 128), meas.unsqueeze(1).expand(B, 4, 128)], -1).view(B * 4, 1024)
    assert torch.equal(hin, want)
    monkeypatch.setattr(D, "_FUSED_CONCAT", True)
    test_fusion_and_decoder_match_oracle(2, False, monkeypatch)
