"""The training iteration: forward_train + reverse sweep of the tape against the REFERENCE's own backward
(golden F13: gradient norm + 8 sampled entries of every live parameter, the names of the dead ones), then the all-reduce /
clip / AdamW half of the step."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(mode, fname="f13_train_gradients_b2.npz"):
    from thinktwice_amd import model as tm, params, synth
    pack = np.load(os.path.join(os.path.dirname(__file__), "golden", fname))
    B, H, W, npts, seed = (int(v) for v in pack["meta"])
    dtype = torch.float32 if mode == "f32" else "f32x3"
    m, cfg = tm.build_thinktwice(final_dim=(H, W), dtype=dtype)
    sd = params.init_params(cfg, seed=seed)
    batch = synth.make_batch(B, img_hw=(H, W), num_points=npts)
    batch.update(synth.make_train_targets(B, img_hw=(H, W)))
    return pack, m, sd, batch


# Bounds: relative error of the per-parameter gradient NORM and of the 8 sampled entries (relative to the norm).  The exact-f32
# mode differs from the reference only by summation order and by the folded-BatchNorm rounding that flips a few ReLU masks on
# these small maps (see tests/test_backward.py); the bf16x3 mode adds its ~1e-5 product error to every layer.
# Round 4 (VERDICT r3 weak #1c): the bounds sit at the measured level now (profiles/r04_pytest_gpu notes): exact f32 worst
# 1.8e-3 (99.9 % of the 878 parameters within 1e-3, 99th percentile 4.3e-4, median 5.9e-6); bf16x3 worst 6.0e-3 (93 % within
# 1e-3).  The parameters beyond 1e-3 in f32 are BatchNorm biases of the LiDAR encoder's first level and GRU biases on the
# 21 x 21 BEV: sums over few, small maps where one flipped ReLU mask is a visible share of the sum.
@pytest.mark.parametrize("mode,tol,fname", [("f32", 2.5e-3, "f13_train_gradients_b2.npz"),
                                            ("f32x3", 1e-2, "f13_train_gradients_b2.npz"),
                                            # F13b: the thinktwice.py size (B=1, 448x896, 65536 points), where one flipped
                                            # ReLU mask is a negligible share of a channel's gradient sum
                                            ("f32", 5e-3, "f13b_train_gradients_fullsize_b1.npz"),
                                            ("f32x3", 1e-2, "f13b_train_gradients_fullsize_b1.npz")],
                         ids=["f13-f32", "f13-bf16x3", "f13b-fullsize-f32", "f13b-fullsize-bf16x3"])
def test_training_backward_matches_reference_gradients_golden_f13(mode, tol, fname, monkeypatch):
    from thinktwice_amd import ops
    from thinktwice_amd.trainer import Trainer
    pack, m, sd, batch = _setup(mode, fname)
    tr = Trainer(m, sd)
    out = tr.backward(batch)
    torch.cuda.synchronize()
    want_total = float(pack["total_loss"][0])
    assert abs(float(out["loss"]) - want_total) < (1e-3 if mode == "f32" else 2e-3) * abs(want_total)
    live = [str(n) for n in pack["names"]]
    missing = [n for n in live if n not in tr.param_grads]
    assert not missing, (len(missing), missing[:10])
    dead = sorted(str(k) for k in pack["dead"])
    got_dead = sorted(k for k in tr.names if k not in tr.param_grads or float(tr.sd[k].grad.abs().max()) == 0.0)
    assert got_dead == dead, (sorted(set(got_dead) ^ set(dead))[:10])
    norm_err, samp_err = {}, {}
    for name, norm, idx, smp in zip(live, pack["norms"], pack["idx"], pack["samples"]):
        g = tr.sd[name].grad.detach().cpu()
        norm = float(norm)
        norm_err[name] = abs(float(g.norm()) - norm) / max(norm, 1e-12)
        got = g.reshape(-1)[torch.from_numpy(idx)].numpy()
        samp_err[name] = float(np.abs(got - smp).max()) / max(norm, 1e-12)
    wn = sorted(norm_err.items(), key=lambda kv: -kv[1])[:5]
    ws = sorted(samp_err.items(), key=lambda kv: -kv[1])[:5]
    ne = np.array(list(norm_err.values()))
    print(mode, "params", len(live), "worst norm rel", wn[0], "worst sample/norm", ws[0],
          "median norm rel", float(np.median(ne)), "share of parameters within 1e-3:", float((ne < 1e-3).mean()),
          "99th percentile", float(np.quantile(ne, 0.99)))
    assert wn[0][1] < tol, wn
    assert ws[0][1] < tol, ws
    if "fullsize" not in fname:
        # north_star's 1e-3 holds for all but a handful of parameters in exact f32 (measured 99.9 %), 93 % in bf16x3
        assert float((ne < 1e-3).mean()) >= (0.995 if mode == "f32" else 0.90), float((ne < 1e-3).mean())
        assert float(np.median(ne)) < (5e-5 if mode == "f32" else 5e-4)
    if "fullsize" in fname:
        # at the thinktwice.py size the camera / LiDAR encoders (most parameters) see maps of 10^3..10^5 pixels: >= 99 % of
        # all gradient norms are within 1e-3 in exact f32 (measured 93.6 % in bf16x3, whose product error of ~1e-5 per layer
        # accumulates through the 50+ layers of the reverse sweep); what is left are the decoder's layers on the 21 x 21 BEV /
        # 2 x 2 flatten maps, where a flipped ReLU mask still shows, whatever the image size
        assert float((ne < 1e-3).mean()) >= (0.99 if mode == "f32" else 0.90), float((ne < 1e-3).mean())


def test_training_step_updates_parameters_and_reduces_the_loss():
    """Trainer.step = backward + (single-rank) all-reduce + global-norm clip + AdamW + operand re-preparation: the update equals
    torch's AdamW on the same gradients, dead parameters stay put exactly (no weight decay: grad is None in torch), and a few iterations on one batch lower
    the loss."""
    from thinktwice_amd.trainer import Trainer
    pack, m, sd, batch = _setup("f32x3")
    tr = Trainer(m, sd, lr=2e-4, weight_decay=1e-7, max_grad_norm=100.0)
    p0 = tr.flat_param.clone()
    out0 = tr.step(batch)
    torch.cuda.synchronize()
    g = tr.grads.flat.clone()
    norm, clip = (float(v) for v in out0["grad_norm"].cpu())
    assert abs(norm - float(g.norm())) < 1e-3 * norm and 0.0 < clip <= 1.0
    # first AdamW step with bias correction: p - lr * (g / (|g| + eps) + wd * p)
    gc = g * clip
    want = p0 - 2e-4 * (gc / (gc.abs() + 1e-8) + 1e-7 * p0)
    live = torch.zeros_like(p0, dtype=torch.bool)
    for off, cnt in tr.live_ranges:
        live[off:off + cnt] = True
    assert 0 < int((~live).sum()) < live.numel() // 10            # the reference's dead parameters are a small minority
    want = torch.where(live, want, p0)                             # ... and torch's AdamW leaves them alone (grad is None)
    assert float((tr.flat_param - want).abs().max()) < 1e-6
    losses = [float(out0["loss"])]
    for _ in range(3):
        losses.append(float(tr.step(batch)["loss"]))
    print("losses over 4 iterations on one batch:", losses)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    # the re-prepared model runs inference with the updated weights
    pred = m.forward_inference(batch)
    assert torch.isfinite(pred["pred_wp"]).all()


def _rank_step(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from thinktwice_amd import model as tm, params, synth
        from thinktwice_amd.trainer import Trainer
        hw = (128, 256)
        m, cfg = tm.build_thinktwice(final_dim=hw, dtype="f32x3")
        tr = Trainer(m, params.init_params(cfg, seed=0), lr=1e-4)
        batch = synth.make_batch(1, img_hw=hw, num_points=4096, seed=100 + rank)      # every rank its own sample
        batch.update(synth.make_train_targets(1, img_hw=hw, seed=200 + rank))
        out = tr.backward(batch)
        local = tr.grads.flat[::997].cpu()
        tr.grads.all_reduce_mean()
        reduced = tr.grads.flat[::997].cpu()
        tr.opt.step(live_ranges=tr.live_ranges)
        torch.cuda.synchronize()
        # (numpy arrays: plain pickles -- torch tensors would travel as shared-memory handles of a process that may be gone)
        q.put((rank, float(out["loss"]), out["log_vars"]["loss"], local.numpy(), reduced.numpy(),
               tr.flat_param[::997].cpu().numpy(), float(tr.flat_param.double().sum())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_training_step_gloo():
    """The data-parallel iteration with world_size 2 (both ranks on this box's one GPU, gloo instead of RCCL): every rank
    runs forward_train + backward on ITS sample, the flat gradient buffer is all-reduced to the mean once, and the AdamW
    step leaves bit-identical parameters on both ranks (apis/mmdet_train.py:67-79: MMDistributedDataParallel)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_step, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, loss0, logged0, g0, red0, p0, sum0), (_, loss1, logged1, g1, red1, p1, sum1) = got
    assert abs(loss0 - loss1) > 1e-3                                   # different samples
    assert abs(logged0 - 0.5 * (loss0 + loss1)) < 1e-4 * abs(logged0)  # _parse_losses logs the all-reduced mean (EDF:431-437)
    assert logged0 == logged1
    assert float(np.abs(g0 - g1).max()) > 0
    want = 0.5 * g0 + 0.5 * g1
    assert np.array_equal(red0, red1) and float(np.abs(red0 - want).max()) <= 1e-6 * float(np.abs(want).max())
    assert np.array_equal(p0, p1) and sum0 == sum1


# ----------------------------------------------------------------------------------------------- model.train() semantics
def _reference_dropout_masks(B, cfg, hw, rng):
    """The keep-masks the reference's two ASPP Dropout(0.5) calls draw (key sweep first, then the older sweep, lss.py:689-714)
    after torch.manual_seed(rng): torch's CPU generator is the same here and where the goldens were made.  Returned as ONE
    channel-last mask in this package's sweep-major image order."""
    N = 4
    mid = cfg["img_encoder"]["depth_net_conf"]["mid_channels"]
    h, w = hw[0] // 16, hw[1] // 16
    torch.manual_seed(rng)
    masks = [torch.nn.functional.dropout(torch.ones(B * N, mid, h, w), 0.5, True) > 0 for _ in range(2)]
    return torch.cat(masks, 0).permute(0, 2, 3, 1).contiguous().to(torch.uint8)


def _setup_train_mode(mode, which):
    """Model, weights and the (calibration-jittered, see synth.make_img_metas) batch of a train-mode golden:
    `which` = "f11" (losses, B = 2) or "f16" (gradients, B = 4)."""
    from thinktwice_amd import model as tm, params, synth
    gold = os.path.join(os.path.dirname(__file__), "golden")
    if which == "f11":
        pack = np.load(os.path.join(gold, "f11_train_losses_trainmode_b2.npz"))
        B, H, W, npts, seed, rng, jitter = (int(v) for v in pack["meta"])
    else:
        pack = np.load(os.path.join(gold, "f16_train_gradients_trainmode_b4.npz"))
        B, H, W, npts, seed = (int(v) for v in pack["meta"])
        rng, jitter = int(pack["rng"][0]), int(pack["calib_jitter"][0])
    m, cfg = tm.build_thinktwice(final_dim=(H, W), dtype=torch.float32 if mode == "f32" else "f32x3")
    sd = params.init_params(cfg, seed=seed)
    batch = synth.make_batch(B, img_hw=(H, W), num_points=npts, jitter_calib=jitter)
    batch.update(synth.make_train_targets(B, img_hw=(H, W)))
    return pack, m, cfg, sd, batch, (B, (H, W), rng)


# Bounds: 1e-3 in the exact-f32 mode.  The bf16x3 mode's ~1e-5 product error is amplified by the batch-statistics BatchNorm1d
# over the B = 2 rows of this golden (output_fc.2: the normalised value is +-1 / sqrt(1 + 4 eps / d^2), ill-conditioned wherever the
# two samples nearly agree): measured 1.05e-3 on speed_loss, <= 1.4e-4 on the other 22 terms.
@pytest.mark.parametrize("mode,tol", [("f32", 1e-3), ("f32x3", 3e-3)])
def test_forward_train_in_train_mode_matches_reference_golden_f11(mode, tol):
    """model.train(): batch-statistics BatchNorm (per sweep in the camera trunk) + the ASPP dropout with the reference's
    masks -> all 23 loss terms against the reference's own forward_train under model.train() (golden F11)."""
    from thinktwice_amd import ops
    f11, m, cfg, sd, batch, (B, hw, rng) = _setup_train_mode(mode, "f11")
    m.load_state_dict(sd)
    m.train()
    ops.DROPOUT_MASKS = iter([_reference_dropout_masks(B, cfg, hw, rng)])
    try:
        losses = m.forward_train(batch)
    finally:
        ops.DROPOUT_MASKS = None
        m.eval()
    torch.cuda.synchronize()
    worst = {}
    for k in f11.files:
        if k in ("meta", "oracle_vs_reference_worst_rel_err"):
            continue
        want = torch.from_numpy(f11[k]).float()
        got = losses[k].detach().float().cpu().reshape(want.shape)
        worst[k] = float((got - want).abs().max() / want.abs().max().clamp_min(1e-6))
    print(mode, "train-mode losses: worst", sorted(worst.items(), key=lambda kv: -kv[1])[:4])
    assert max(worst.values()) < tol, sorted(worst.items(), key=lambda kv: -kv[1])[:6]
    assert sorted(worst.values())[-2] < 1e-3          # every term but the worst one inside 1e-3 in both modes


@pytest.mark.parametrize("mode", ["f32", "f32x3"])
def test_training_backward_in_train_mode_matches_reference_gradients_golden_f16(mode, monkeypatch):
    """The whole iteration's gradients under model.train() against the reference's loss.backward() under model.train()
    (golden F16, B = 4): total loss, the same live / dead parameter sets, per-parameter gradient norms and sampled entries.

    Bounds.  Under batch statistics two f32 implementations agree to ~1e-6 on the forward (the statistics are summed in a
    different order) instead of ~1e-7, and every BatchNorm centres its pre-activations on zero: on this model's small maps
    (21x21 ... 2x2 BEV maps, B x 4 future maps per refinement stage) a handful of ReLU masks flip, each moving the gradients of
    its block by ~1e-2 and of everything upstream of it (both encoders: most parameters) by a few 1e-3.  Every building block is
    exact against autograd on its own (tests/test_batchnorm.py, tests/test_backward.py train variants, tools/debug_*.py:
    5e-7 .. 1e-5); this test pins the composition: median <= 2e-2, 90th percentile <= 6e-2, worst <= 0.25."""
    from thinktwice_amd import ops
    from thinktwice_amd.trainer import Trainer
    pack, m, cfg, sd, batch, (B, hw, rng) = _setup_train_mode(mode, "f16")
    tr = Trainer(m, sd, frozen_bn=False)
    ops.DROPOUT_MASKS = iter([_reference_dropout_masks(B, cfg, hw, rng)])
    try:
        out = tr.backward(batch)
    finally:
        ops.DROPOUT_MASKS = None
    torch.cuda.synchronize()
    want_total = float(pack["total_loss"][0])
    assert abs(float(out["loss"]) - want_total) < 2e-3 * abs(want_total), (float(out["loss"]), want_total)
    live = [str(n) for n in pack["names"]]
    missing = [n for n in live if n not in tr.param_grads]
    assert not missing, (len(missing), missing[:10])
    dead = sorted(str(k) for k in pack["dead"])
    got_dead = sorted(k for k in tr.names if k not in tr.param_grads or float(tr.sd[k].grad.abs().max()) == 0.0)
    assert got_dead == dead, (sorted(set(got_dead) ^ set(dead))[:10])
    norm_err, samp_err = {}, {}
    noise = set(str(n) for n in pack["noise"])
    for name, norm, idx, smp in zip(live, pack["norms"], pack["idx"], pack["samples"]):
        g = tr.sd[name].grad.detach().cpu()
        norm = float(norm)
        if name in noise:                # (conv biases in front of a batch-statistics BN: analytically zero gradients)
            assert float(g.norm()) < 5e-3, (name, float(g.norm()))
            continue
        norm_err[name] = abs(float(g.norm()) - norm) / norm
        got = g.reshape(-1)[torch.from_numpy(idx)].numpy()
        samp_err[name] = float(np.abs(got - smp).max()) / norm
    wn = sorted(norm_err.items(), key=lambda kv: -kv[1])[:5]
    ws = sorted(samp_err.items(), key=lambda kv: -kv[1])[:5]
    ne = np.array(list(norm_err.values()))
    print(mode, "train-mode params", len(live), "worst norm rel", wn[0], "worst sample/norm", ws[0],
          "median norm rel", float(np.median(ne)), "p90", float(np.quantile(ne, 0.9)))
    assert float(np.median(ne)) < 2e-2 and float(np.quantile(ne, 0.9)) < 6e-2 and wn[0][1] < 0.25, wn
    assert ws[0][1] < 0.25, ws
    # running statistics moved (momentum update), parameters did not (backward only)
    k = "img_encoder.img_backbone.bn1.running_mean"
    assert float((tr.buffers[k].cpu() - sd[k]).abs().max()) > 0


def test_two_trainers_in_one_process_do_not_disturb_each_other():
    """a second Trainer preparing (and re-preparing) its operands must leave the first one's tape
    metadata alone -- the first model's backward gives the same gradients before and after (to the f32 atomics of the three backward scatters)."""
    from thinktwice_amd import model as tm, params, synth
    from thinktwice_amd.trainer import Trainer
    hw = (128, 256)
    ma, cfg = tm.build_thinktwice(final_dim=hw, dtype="f32x3")
    ta = Trainer(ma, params.init_params(cfg, seed=0), lr=1e-4)
    batch = tm.batch_to_device(synth.make_batch(1, img_hw=hw, num_points=4096, seed=3))
    batch.update(synth.make_train_targets(1, img_hw=hw, seed=4))
    ta.backward(batch)
    g0 = ta.grads.flat.clone()
    n_a = len(ta.param_grads)
    mb, _ = tm.build_thinktwice(final_dim=hw, dtype="f32x3")
    tb = Trainer(mb, params.init_params(cfg, seed=1), lr=1e-4)          # prepares model B: clears ITS entries only
    tb.step(batch)                                                      # ... and re-prepares them after its update
    ta.backward(batch)                                                  # model A's tape still finds every parameter
    assert len(ta.param_grads) == n_a
    assert float((ta.grads.flat - g0).norm()) <= 1e-4 * float(g0.norm())
    tb.backward(batch)
    assert len(tb.param_grads) == n_a


def test_checkpoint_has_the_torch_layout_and_resumes_bit_identically():
    """`Trainer.checkpoint()` = mmcv's save_checkpoint content: meta (epoch, iter), state_dict (reference names,
    BatchNorm call counters advanced), optimizer in torch.optim.AdamW.state_dict() layout -- loadable by a real torch AdamW
    over parameters of the same shapes -- and a resumed Trainer continues exactly where the saved one was."""
    from thinktwice_amd import model as tm, params, synth
    from thinktwice_amd.trainer import Trainer
    hw = (128, 256)
    m, cfg = tm.build_thinktwice(final_dim=hw, dtype="f32x3")
    sd0 = params.init_params(cfg, seed=0)
    tr = Trainer(m, sd0, lr=1e-4, frozen_bn=False)
    batch = tm.batch_to_device(synth.make_batch(2, img_hw=hw, num_points=4096, seed=3))
    batch.update(synth.make_train_targets(2, img_hw=hw, seed=4))
    tr.step(batch)
    tr.step(batch)
    ck = tr.checkpoint(epoch=3)
    assert ck["meta"] == {"epoch": 3, "iter": 2}
    osd = ck["optimizer"]
    assert set(osd) == {"state", "param_groups"} and osd["param_groups"][0]["params"] == list(range(len(tr.names)))
    live = [i for i, k in enumerate(tr.names) if k in tr.param_grads]
    assert sorted(osd["state"]) == live and len(live) == 878                  # the 90 dead parameters carry no state (torch)
    i0 = live[0]
    assert osd["state"][i0]["exp_avg"].shape == tr.sd[tr.names[i0]].shape and float(osd["state"][i0]["step"]) == 2.0
    # a real torch AdamW over same-shaped parameters accepts it
    ps = [torch.nn.Parameter(torch.zeros(tr.sd[k].shape)) for k in tr.names]
    opt = torch.optim.AdamW(ps, lr=1e-4)
    opt.load_state_dict(osd)
    assert torch.equal(opt.state[ps[i0]]["exp_avg"], osd["state"][i0]["exp_avg"])
    # BatchNorm call counters: T = 2 calls per iteration inside the per-sweep camera pass, 1 elsewhere
    k_cam = "img_encoder.img_backbone.bn1.num_batches_tracked"
    k_lid = next(k for k in sd0 if k.startswith("lidar_encoder.") and k.endswith("num_batches_tracked"))
    assert int(ck["state_dict"][k_cam]) == int(sd0[k_cam]) + 4 and int(ck["state_dict"][k_lid]) == int(sd0[k_lid]) + 2
    # resume: a fresh Trainer loaded from the checkpoint takes the same third step
    m2, _ = tm.build_thinktwice(final_dim=hw, dtype="f32x3")
    tr2 = Trainer(m2, params.init_params(cfg, seed=1), lr=1e-4, frozen_bn=False)
    tr2.load_checkpoint(ck)
    assert (tr2.iteration, tr2.epoch, tr2.opt.steps) == (2, 3, 2)
    assert torch.equal(tr2.flat_param, tr.flat_param) and torch.equal(tr2.opt.m, tr.opt.m) and torch.equal(tr2.opt.v, tr.opt.v)
    for k, b in tr.buffers.items():
        assert torch.equal(tr2.buffers[k], b), k


def test_non_finite_gradient_norm_skips_the_update_on_the_device():
    """the finite check used to run AFTER AdamW had written NaNs into the weights.  tt_grad_norm_clip now hands the
    update a NaN factor for a non-finite norm and tt_adamw_step leaves p / m / v untouched."""
    from thinktwice_amd.optim import FlatAdamW
    n = 100_000
    p = torch.randn(n, device="cuda")
    g = torch.randn(n, device="cuda")
    opt = FlatAdamW(p, g, lr=1e-3)
    opt.step()
    p1, m1, v1 = p.clone(), opt.m.clone(), opt.v.clone()
    for poison in (float("nan"), float("inf")):
        g.normal_()
        g[12345] = poison
        ns = opt.step().cpu()
        assert not torch.isfinite(ns[0]) and ns[1] != ns[1]
        assert torch.equal(p, p1) and torch.equal(opt.m, m1) and torch.equal(opt.v, v1)
    # the step COUNT stays with the moments (it lives on the device): two skipped iterations later the applied
    # count is still 1, the next good update uses the bias corrections of step 2, exactly like a torch AdamW that never saw
    # the two bad iterations
    assert opt.steps == 1 and opt.issued == 3 and opt.skipped() == 2
    ref_p = torch.nn.Parameter(p1.cpu().clone())
    ref = torch.optim.AdamW([ref_p], lr=1e-3, weight_decay=1e-7)
    ref.state[ref_p] = {"step": torch.tensor(1.0), "exp_avg": m1.cpu().clone(), "exp_avg_sq": v1.cpu().clone()}
    g.normal_()
    ref_p.grad = g.cpu().clone()
    torch.nn.utils.clip_grad_norm_([ref_p], 100.0)
    ref.step()
    opt.step()
    assert opt.steps == 2 and opt.skipped() == 2
    assert not torch.equal(p, p1) and torch.isfinite(p).all()
    assert float((p.cpu() - ref_p.detach()).abs().max()) < 2e-6
    sd = opt.state_dict()
    assert sd["step"] == 2


def test_resume_from_an_mmcv_style_optimizer_dict_takes_the_base_rate_from_initial_lr():
    """an mmcv checkpoint's param_group carries the BASE rate as 'initial_lr' and the already scheduled rate of the
    save iteration as 'lr'.  Resuming past the warm-up must anneal from the base rate -- not apply the schedule to the scheduled
    rate a second time -- and the Trainer's own checkpoints must say the same thing."""
    from thinktwice_amd import model as tm, params
    from thinktwice_amd.optim import warmup_cosine_lr
    from thinktwice_amd.trainer import Trainer
    hw = (128, 256)
    m, cfg = tm.build_thinktwice(final_dim=hw, dtype="f32x3")
    tr = Trainer(m, params.init_params(cfg, seed=0), lr=5e-4, frozen_bn=True)
    sched = dict(total_iters=6000, iters_per_epoch=100, warmup_iters=1000, warmup_ratio=1.0 / 3, min_lr_ratio=1e-3)
    tr.set_schedule(**sched)
    it = 2500                                                    # past the warm-up, in epoch 25 of 60
    base = 2e-4
    scheduled = warmup_cosine_lr(base, it, **sched)
    assert scheduled < 0.9 * base
    osd = {"state": {}, "param_groups": [{"lr": scheduled, "initial_lr": base, "betas": (0.9, 0.999), "eps": 1e-8,
                                          "weight_decay": 0.01, "params": list(range(len(tr.names)))}]}
    tr.load_checkpoint({"meta": {"epoch": 25, "iter": it}, "state_dict": tr.state_dict(), "optimizer": osd})
    assert tr.opt.lr == base
    assert abs(tr.current_lr() - scheduled) <= 1e-12 * base
    g = tr.optimizer_state_dict()["param_groups"][0]
    assert g["initial_lr"] == base and abs(g["lr"] - scheduled) <= 1e-12 * base
    # a checkpoint without an optimizer entry: updates skipped BEFORE the load are not charged to the loaded iteration count
    tr.opt.issued += 1                                           # an issued update the device did not apply
    assert tr.opt.skipped() == 1
    tr.load_checkpoint({"meta": {"epoch": 1, "iter": 120}, "state_dict": tr.state_dict()})
    assert tr.reconcile() == 0 and tr.iteration == 120
