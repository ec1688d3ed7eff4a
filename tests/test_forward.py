"""End-to-end `forward_inference` (BASELINE configs 1-3): HIP model vs the oracle (reduced image size,
seconds on CPU) and vs the committed golden vectors produced by the REFERENCE modules
(f7: B=2 128x256; f8: B=1 full 448x896 thinktwice.py size; f14: B=8 full size = BASELINE configs 2 / 3).
Tolerance 1e-3 of each tensor's max (north_star) for the f32 mode and for the 16-bit headline mode (IEEE half
storage) on the 14 output keys; the bf16 mode and the trunk intermediates of the half mode carry their measured
bounds."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KEYS = ("pred_wp", "mu_branches", "sigma_branches", "future_mu", "future_sigma", "pred_speed", "pred_value_traj",
        "pred_value_ctrl", "pred_features_traj", "pred_features_ctrl", "bev_feature", "refine_BEV_feature",
        "refine_flattned_BEV_feature", "refine_future_BEV_feature")


def _check_against_pack(pack, out, tol):
    errs = {}
    for k in KEYS:
        v = out[k].detach().float().cpu()
        if k in pack.files:
            want = torch.from_numpy(pack[k])
            assert v.shape == want.shape, k
            errs[k] = float((v - want).abs().max() / want.abs().max().clamp_min(1e-6))
        else:
            idx = torch.from_numpy(pack[k + "__idx"])
            want = torch.from_numpy(pack[k + "__val"])
            errs[k] = float((v.reshape(-1)[idx] - want).abs().max() / float(pack[k + "__stats"][2]))
    assert max(errs.values()) < tol, errs
    return errs


def _run_model(B, hw, npts, seed, dtype=torch.float32):
    from thinktwice_amd import model as tm, params, synth
    m, cfg = tm.build_thinktwice(dtype=dtype, final_dim=hw)
    sd = params.init_params(cfg, seed=seed)
    m.load_state_dict(sd)
    batch = synth.make_batch(B, img_hw=hw, num_points=npts)
    out = m.forward_inference(tm.batch_to_device(batch))
    torch.cuda.synchronize()
    return out, cfg, sd, batch


def _check_look_counts(out, pack, strict, tag=""):
    """Integer work of the look module against the reference: per layer the number of queries whose projection lands inside
    each camera image (`look_count` [layer][sample][camera]) and the padded length `max_len`.  `strict`: bit-equal (the exact
    f32 mode AND the bf16x3 headline mode -- VERDICT r3 weak #1b); otherwise the flip rate is reported, not asserted."""
    flips, total = 0, 0
    for L in range(5):
        got = out["_look_info"][L][0].cpu().numpy().reshape(pack["look_count"][L].shape)
        want = pack["look_count"][L]
        flips += int(np.abs(got.astype(np.int64) - want.astype(np.int64)).sum())
        total += int(want.sum())
        if strict:
            np.testing.assert_array_equal(got, want, err_msg=f"{tag} layer {L}: in-image query counts differ from the reference")
            assert int(out["_look_info"][L][1].item()) == int(pack["look_max_len"][L]), (tag, L)
    print(f"{tag} look-module in-image tests: {flips} flipped of {total} hits ({'asserted equal' if strict else 'reported only'})")
    return flips, total


def test_forward_small_matches_reference_golden_and_oracle(golden_dir):
    from oracle import model_ref as M
    pack = np.load(os.path.join(golden_dir, "f7_forward_small_b2.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    out, cfg, sd, batch = _run_model(B, (H, W), npts, seed)
    errs = _check_against_pack(pack, out, 1e-3)
    print("f7 (reference golden) rel errs", errs)
    with torch.no_grad():
        ref = M.forward_inference(sd, cfg, batch, return_intermediates=True)
    for k in KEYS:
        e = float((out[k].cpu() - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-6))
        assert e < 1e-3, (k, e)
    _check_look_counts(out, pack, strict=True, tag="f7 f32")
    # waypoint L2 vs reference (BASELINE metric "waypoint L2 vs ref")
    l2 = float((out["pred_wp"].cpu() - torch.from_numpy(pack["pred_wp"])).norm(dim=-1).max())
    assert l2 < 1e-3, l2


@pytest.mark.parametrize("dt", [torch.float32, "f32x3", "f32x3h"], ids=["f32", "bf16x3", "bf16x3h"])
def test_forward_is_bit_reproducible(dt):
    """Two forwards of the same batch are torch.equal in every output: no floating-point atomics on the inference path
    (ordered split-K slices, lift-splat partial rows reduced in strip order)."""
    outs = []
    for _ in range(2):
        out, _, _, _ = _run_model(2, (128, 256), 3000, 11, dtype=dt)
        outs.append({k: out[k].clone() for k in KEYS})
    for k in KEYS:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_forward_full_size_matches_reference_golden(golden_dir):
    pack = np.load(os.path.join(golden_dir, "f8_forward_full_b1.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    assert (H, W) == (448, 896)
    out, *_ = _run_model(B, (H, W), npts, seed)
    errs = _check_against_pack(pack, out, 1e-3)
    print("f8 (reference golden, thinktwice.py size) rel errs", errs)
    assert out["pred_wp"].shape == (1, 6, 4, 2)
    _check_look_counts(out, pack, strict=True, tag="f8 f32")


@pytest.mark.parametrize("dt", ["f32x3", "f32x3h"], ids=["bf16x3", "bf16x3h"])
def test_forward_full_size_headline_mode_integer_parity(golden_dir, dt):
    """F8 in the all-bf16x3 mode and in the headline mode (bf16x3 + the PAFPN's 3 x 3 layers on the h2 product): the 14 outputs
    within 1e-3 AND the look module's integer work bit-equal."""
    pack = np.load(os.path.join(golden_dir, "f8_forward_full_b1.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    out, *_ = _run_model(B, (H, W), npts, seed, dtype=dt)
    errs = _check_against_pack(pack, out, 1e-3)
    print(f"f8 {dt}: rel errs", errs, "inter", _inter_errs(pack, out) if "inter__seg__idx" in pack.files else None)
    _check_look_counts(out, pack, strict=True, tag=f"f8 {dt}")


def _inter_errs(pack, out):
    """Trunk intermediates of the golden packs (sampled values of the oracle, which equals the reference bit for bit
    on every module output): segmentation logits (config 3 "waypoint/seg outputs") and the camera BEV (config 2)."""
    seg = out["_seg_cl"][..., :12].permute(0, 3, 1, 2).float().cpu().reshape(-1)
    bev = out["_cam_bev_cl"].permute(0, 3, 1, 2).float().cpu().reshape(-1)
    errs = {}
    for name, v in (("seg", seg), ("cam_bev", bev)):
        idx = torch.from_numpy(pack[f"inter__{name}__idx"])
        want = torch.from_numpy(pack[f"inter__{name}__val"])
        errs[name] = float((v[idx] - want).abs().max() / float(pack[f"inter__{name}__stats"][2]))
    return errs


# (precision mode, bound on the 14 output keys, bound on seg / camera BEV).  The exact-f32 mode and the bf16x3 mode (f32
# storage, three bf16 MFMAs per product) meet the 1e-3 tolerance on every output of forward_inference; the two 16-bit
# STORAGE modes are held to their measured level (gpurun_out/r2_pytest_b_model.log, B=8): bf16, 8 mantissa bits ->
# pred_wp 1.8e-2, value head 6.3e-2, seg 9.7e-3; IEEE half, 11 bits -> pred_wp 2.3e-3, value head 9.6e-3, seg 1.2e-3:
# 8x below bf16 but not inside 1e-3.  bf16x3 measures pred_wp 3.2e-5, worst key 1.2e-4.
# "f32x3h" (round 6, the bench headline): bf16x3 with the PAFPN's ten 3 x 3 convolutions on the two-MFMA h2 product -- each reads an
# IEEE-half COPY of its input (f16 (hi, lo) weights), the neck's sums stay f32.  The one stage the storage-level emulation clears
# (profiles/r06_precision_mix_storage.txt; with the sums themselves in half the waypoint distance was 1.19 mm, with the sums in f32
# it is 0.86 mm: profiles/r06_x3h_mode.txt).  Held to the same bounds as the other parity modes.
MODES = [(torch.float32, 1e-3, 1e-3), ("f32x3", 1e-3, 1e-3), ("f32x3h", 1e-3, 1e-3), (torch.float16, 2e-2, 5e-3),
         (torch.bfloat16, 0.15, 4e-2)]
# pred_wp (relative to its max) and the waypoint L2 distance in metres (BASELINE metric "waypoint L2 vs ref")
WP_TOL = {torch.float32: (1e-3, 1e-3), "f32x3": (1e-3, 1e-3), "f32x3h": (1e-3, 1e-3), torch.float16: (5e-3, 3e-2),
          torch.bfloat16: (4e-2, 0.25)}


@pytest.mark.parametrize("dt,tol,tol_inter", MODES, ids=["f32", "bf16x3", "bf16x3h", "f16", "bf16"])
def test_forward_batch8_full_size_matches_reference_golden(golden_dir, dt, tol, tol_inter):
    """BASELINE configs 2 / 3: batch 8 at the thinktwice.py size against the REFERENCE's forward_inference (f14).  The
    SCA batch coupling (multi_scale_deformable_attn_function.py:338-341: first `bs` slots zeroed, divide by `bs`)
    makes B=8 different arithmetic from the B=1/2 goldens."""
    pack = np.load(os.path.join(golden_dir, "f14_forward_full_b8.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    assert (B, H, W) == (8, 448, 896)
    from thinktwice_amd import model as tm, params, synth
    m, cfg = tm.build_thinktwice(dtype=dt, final_dim=(H, W))
    m.load_state_dict(params.init_params(cfg, seed=seed))
    batch = synth.make_batch(B, img_hw=(H, W), num_points=npts)
    out = m.forward_inference(tm.batch_to_device(batch))
    torch.cuda.synchronize()
    errs = _check_against_pack(pack, out, tol)
    inter = _inter_errs(pack, out)
    l2 = float((out["pred_wp"].cpu() - torch.from_numpy(pack["pred_wp"])).norm(dim=-1).max())
    print(f"f14 B=8 {dt}: rel errs", errs, "inter", inter, "waypoint L2 max", l2)
    assert max(inter.values()) < tol_inter, inter
    assert errs["pred_wp"] < WP_TOL[dt][0] and l2 < WP_TOL[dt][1], (errs["pred_wp"], l2)
    # integer work bit-exact in the exact-f32 mode and in the bf16x3 HEADLINE mode (a 16-bit STORAGE trunk may move a
    # projected waypoint across an image edge: there the flip rate is printed)
    _check_look_counts(out, pack, strict=dt in (torch.float32, "f32x3", "f32x3h"), tag=f"f14 {dt}")


@pytest.mark.parametrize("dt,tol,tol_inter", MODES[1:], ids=["bf16x3", "bf16x3h", "f16", "bf16"])
def test_forward_16bit_modes_small(golden_dir, dt, tol, tol_inter):
    pack = np.load(os.path.join(golden_dir, "f7_forward_small_b2.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    out, *_ = _run_model(B, (H, W), npts, seed, dtype=dt)
    errs = _check_against_pack(pack, out, tol)
    print(f"{dt} trunk: rel errs vs reference golden", errs)
    _check_look_counts(out, pack, strict=dt in ("f32x3", "f32x3h"), tag=f"f7 {dt}")


def test_mmcv_style_checkpoint_loading_through_the_module_shell(golden_dir):
    """B2: EncoderDecoder is an nn.Module shell; a checkpoint reaches it the way mmcv's load_checkpoint / load_state_dict
    deliver it (mmcv/runner/checkpoint.py: strip the DataParallel `module.` prefix, then walk the module tree calling
    `_load_from_state_dict`), and `state_dict()` hands the same reference-format dict back."""
    from thinktwice_amd import model as tm, params, synth
    pack = np.load(os.path.join(golden_dir, "f7_forward_small_b2.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    m, cfg = tm.build_thinktwice(final_dim=(H, W))
    assert isinstance(m, torch.nn.Module)
    sd = params.init_params(cfg, seed=seed)
    ckpt = {"state_dict": {"module." + k: v for k, v in sd.items()}, "meta": {}}

    def mmcv_load_state_dict(module, state_dict):        # the loader's tree walk, restated
        state_dict = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        missing, unexpected, errs = [], [], []

        def load(mod, prefix=""):
            mod._load_from_state_dict(state_dict, prefix, {}, True, missing, unexpected, errs)
            for name, child in mod._modules.items():
                if child is not None:
                    load(child, prefix + name + ".")
        load(module)
        assert not errs
    mmcv_load_state_dict(m, ckpt["state_dict"])
    m.eval()
    out = m.forward_inference(tm.batch_to_device(synth.make_batch(B, img_hw=(H, W), num_points=npts)))
    torch.cuda.synchronize()
    _check_against_pack(pack, out, 1e-3)
    back = m.state_dict()
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in list(sd)[:50])


def test_forward_refuses_missing_weights():
    from thinktwice_amd import _lib, model as tm
    m, _ = tm.build_thinktwice(final_dim=(128, 256))
    with pytest.raises(_lib.TTError):
        m.forward_inference({})


def test_inference_graph_replay_matches_eager_and_golden(golden_dir):
    """The captured HIP graph (bench.py's launch mode) runs the same kernels as the eager forward: its outputs match
    the eager ones (the inference path has had no floating-point atomics since round 3; the bound stays at f32 rounding
    because graph memory places buffers differently), still match the reference golden, and a replay
    after `update()` with different inputs tracks the eager result for those inputs."""
    from thinktwice_amd import model as tm, params, synth
    from thinktwice_amd.encoder_decoder import InferenceGraph
    pack = np.load(os.path.join(golden_dir, "f7_forward_small_b2.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    m, cfg = tm.build_thinktwice(dtype=torch.float32, final_dim=(H, W))
    m.load_state_dict(params.init_params(cfg, seed=seed))
    batch = tm.batch_to_device(synth.make_batch(B, img_hw=(H, W), num_points=npts))
    eager = {k: v.clone() for k, v in m.forward_inference(batch).items() if k in KEYS}
    g = InferenceGraph(m, batch, channel_last_out=False)
    out = g.replay()
    torch.cuda.synchronize()
    _check_against_pack(pack, out, 1e-3)
    for k in KEYS:
        e = float((out[k] - eager[k]).abs().max() / eager[k].abs().max().clamp_min(1e-6))
        assert e < 1e-5, (k, e)
    # new inputs through the static buffers
    batch2 = tm.batch_to_device(synth.make_batch(B, img_hw=(H, W), num_points=npts, seed=77))
    eager2 = {k: v.clone() for k, v in m.forward_inference(batch2).items() if k in KEYS}
    out2 = g(batch2)
    torch.cuda.synchronize()
    for k in KEYS:
        e = float((out2[k] - eager2[k]).abs().max() / eager2[k].abs().max().clamp_min(1e-6))
        assert e < 1e-5, (k, e)
    assert float((out2["pred_wp"] - eager["pred_wp"]).abs().max()) > 0   # the replay really saw the new inputs


def test_prev_sweep_cache_matches_two_sweep_forward():
    """the older sweep is encoded and splatted with the key frame's matrices, so its BEV equals the
    key-sweep BEV of the tick where that image was the key frame.  Feeding the cached BEV (camera trunk on the key
    sweep only) must reproduce the full two-sweep forward; f32 tolerance 1e-4 (the half-size conv batch may take a
    different tile / K order), and the PrevSweepCache tick driver must do the same on its own."""
    from thinktwice_amd import model as tm, params, synth
    from thinktwice_amd.encoder_decoder import PrevSweepCache
    B, hw, npts = 2, (128, 256), 4000
    m, cfg = tm.build_thinktwice(dtype=torch.float32, final_dim=hw)
    m.load_state_dict(params.init_params(cfg, seed=3))
    a = synth.make_batch(B, img_hw=hw, num_points=npts, seed=11)
    b = synth.make_batch(B, img_hw=hw, num_points=npts, seed=29)
    a["img"][:, 1] = b["img"][:, 0]            # tick t-lag: the key frame (last T index) is b's older sweep
    a, b = tm.batch_to_device(a), tm.batch_to_device(b)
    key_a = m.forward_inference(a)["_key_bev_cl"].contiguous().clone()
    full = {k: v.clone() for k, v in m.forward_inference(b).items() if k in KEYS + ("_cam_bev_cl",)}
    cached = m.forward_inference(b, prev_bev=key_a)
    torch.cuda.synchronize()
    for k in KEYS + ("_cam_bev_cl",):
        e = float((cached[k] - full[k]).abs().max() / full[k].abs().max().clamp_min(1e-6))
        assert e < 1e-4, (k, e)
    # ... and the ORACLE's two-sweep forward of tick b (not just this implementation's own): 1e-3
    from oracle import model_ref as M
    sd = params.init_params(cfg, seed=3)
    b_host = synth.make_batch(B, img_hw=hw, num_points=npts, seed=29)
    with torch.no_grad():
        ref = M.forward_inference(sd, cfg, b_host)
    for k in KEYS:
        e = float((cached[k].cpu() - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-6))
        assert e < 1e-3, ("cached tick vs oracle", k, e)
    drv = PrevSweepCache(m, lag=1)
    drv.tick(a)
    t2 = drv.tick(b)
    torch.cuda.synchronize()
    for k in KEYS:
        e = float((t2[k] - full[k]).abs().max() / full[k].abs().max().clamp_min(1e-6))
        assert e < 1e-4, (k, e)
    # and the cache really removes work: a one-sweep image tensor is accepted
    b1 = dict(b)
    b1["img"] = b["img"][:, 1:].contiguous()
    one = m.forward_inference(b1, prev_bev=key_a)
    assert float((one["pred_wp"] - cached["pred_wp"]).abs().max()) < 1e-5      # (same kernels on a one-sweep batch: f32 rounding at most)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_forward_train_losses_match_reference_golden_f10(mode):
    """Row A24 / SURVEY 8f-4 (loss half): `forward_train` = forward + teacher-forcing pass + every loss term as a device
    reduction (csrc/losses.hip), against the reference's own `forward_train` outputs (golden F10, B=2 128x256, running-
    statistics BN); then `train_step` / `model(**batch)` return dict(loss, log_vars, num_samples) with the total of
    `_parse_losses` (EDF:140-145, 409-439)."""
    from thinktwice_amd import model as tm, params, synth
    pack = np.load(os.path.join(os.path.dirname(__file__), "golden", "f10_train_losses_b2.npz"))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    dtype = torch.float32 if mode == "f32" else "f32x3"
    m, cfg = tm.build_thinktwice(final_dim=(H, W), dtype=dtype)
    m.load_state_dict(params.init_params(cfg, seed=seed))
    batch = synth.make_batch(B, img_hw=(H, W), num_points=npts)
    batch.update(synth.make_train_targets(B, img_hw=(H, W)))
    losses = m.forward_train(batch)
    torch.cuda.synchronize()
    names = [k for k in pack.files if k not in ("meta", "oracle_vs_reference_worst_rel_err")]
    assert list(losses.keys()) == names, (list(losses.keys()), names)       # same terms, same order
    tol = 1e-3 if mode == "f32" else 2e-3
    worst = {}
    for k in names:
        got, want = losses[k].detach().cpu().numpy(), pack[k]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        worst[k] = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-6))
    print(mode, "loss rel errs:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) < tol, worst
    from thinktwice_amd.losses import LossReducer                           # the reductions are bit-reproducible
    red, x = LossReducer("cuda"), torch.randn(8, 5, 4, 32, 21, 21, device="cuda")
    t = torch.randn(8, 4, 32, 21, 21, device="cuda")
    assert torch.equal(red.smooth_l1(x, t, clamp_max=5.0, scale=0.25), red.smooth_l1(x, t, clamp_max=5.0, scale=0.25))
    ref = torch.clamp(torch.nn.functional.smooth_l1_loss(x, t.unsqueeze(1).expand_as(x), reduction="none"), max=5.0).mean()
    assert abs(float(red.smooth_l1(x, t, clamp_max=5.0)) - float(ref)) < 1e-6 * float(ref) + 1e-7
    out = m.train_step(batch, None)
    want_total = sum(float(pack[k].mean()) for k in names if "loss" in k)
    assert out["num_samples"] == B and set(out) == {"loss", "log_vars", "num_samples"}
    assert abs(float(out["loss"]) - want_total) / abs(want_total) < tol
    assert abs(out["log_vars"]["loss"] - want_total) / abs(want_total) < tol and set(names) < set(out["log_vars"])
    out2 = m(**batch)                                                       # the mmcv runner's entry (EDF:393-407)
    assert abs(float(out2["loss"]) - float(out["loss"])) < 1e-5 * abs(float(out["loss"]))   # same forward twice


def test_module_shell_init_weights_and_device_placement():
    """B2 boundary (VERDICT r3 missing #2 / weak #7): train.py:225 `model.init_weights()` gives a freshly built model the
    restated initialisation (and leaves a loaded checkpoint alone); `.to()` / `.cuda()` are no-ops only for the device the
    model is on, refuse a dtype / the CPU / a GPU that is not there, and never silently ignore their argument."""
    from thinktwice_amd import _lib, model as tm, params, synth
    hw = (128, 256)
    m, cfg = tm.build_thinktwice(final_dim=hw)
    assert not m.loaded
    assert m.init_weights() is m and m.loaded
    want = params.init_params(cfg, seed=0)
    got = m.state_dict()
    assert list(got.keys()) == list(want.keys())
    for k in ("img_encoder.img_backbone.layer1.0.conv1.weight", "decoder.decoder_layers.0.look_module.img_sca.deformable_attention.sampling_offsets.bias",
              "lidar_encoder.pts_backbone.blocks.0.0.weight"):
        if k in want:
            assert torch.equal(got[k], want[k]), k
    batch = tm.batch_to_device(synth.make_batch(1, img_hw=hw, num_points=4096))
    a = m.forward_inference(batch)["pred_wp"].clone()
    m2, _ = tm.build_thinktwice(final_dim=hw)
    m2.load_state_dict(want)
    assert torch.equal(a, m2.forward_inference(batch)["pred_wp"])
    # a second init_weights() keeps what is loaded (mmcv: weights that came through init_cfg are not re-initialised)
    sd_id = id(m._ref_sd)
    m.init_weights(seed=5)
    assert id(m._ref_sd) == sd_id
    # device placement
    enc = m.img_encoder
    assert m.to("cuda") is m and m.cuda() is m and m.to(torch.device("cuda", 0)) is m and m.to(device="cuda:0") is m
    assert m.img_encoder is enc                                     # nothing was rebuilt
    for bad in (lambda: m.to(torch.float16), lambda: m.to(dtype=torch.bfloat16), lambda: m.cpu(), lambda: m.to("cpu"),
                lambda: m.to(f"cuda:{torch.cuda.device_count()}")):
        with pytest.raises(_lib.TTError):
            bad()
    assert m.img_encoder is enc and m.loaded                        # a refused move leaves the model intact
    assert torch.equal(a, m.forward_inference(batch)["pred_wp"])
    # train() / eval() reach the sub-objects that key on the mode (LiDAR voxel cap: ADVICE r3)
    m.train()
    assert m.training and m.lidar_encoder.training
    m.eval()
    assert not m.training and not m.lidar_encoder.training


def test_forwards_in_flight_on_alternating_streams_are_bit_equal_to_serial():
    """bench.py keeps three batches in flight (consecutive forwards on alternating streams, so that the decoder's per-sample kernels
    run under the next batch's camera trunk): the model holds no per-call state that two forwards in flight could share (the
    bordered image buffer is per stream, the wide chains' ticket slots are claimed per launch), so every forward equals the
    serial one bit for bit."""
    from thinktwice_amd import model as tm, params, synth
    hw = (128, 256)
    m, cfg = tm.build_thinktwice(dtype="f32x3", final_dim=hw)
    m.load_state_dict(params.init_params(cfg, seed=5))
    batches = [tm.batch_to_device(synth.make_batch(2, img_hw=hw, num_points=3000, seed=40 + i)) for i in range(2)]
    serial = []
    for b in batches:
        out = m.forward_inference(b)
        serial.append({k: out[k].clone() for k in KEYS})
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    outs = []
    for i in range(9):
        with torch.cuda.stream(streams[i % 3]):
            outs.append((i % 2, m.forward_inference(batches[i % 2])))
    torch.cuda.synchronize()
    for which, out in outs:
        for k in KEYS:
            assert torch.equal(out[k], serial[which][k]), (which, k)
