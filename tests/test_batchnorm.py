"""Train-mode BatchNorm / dropout kernels (csrc/batchnorm.hip) vs torch: F.batch_norm(training=True) forward
(batch statistics per row group, running-statistics update), autograd backward (dz, residual gradients, dgamma, dbeta), the
sparse form with a device row count, and nn.Dropout's scaling with a replayed mask."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _spec(C, g, eps=1e-5, momentum=0.1):
    from thinktwice_amd import ops
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    spec = ops.BNSpec("layer.bn", gamma.cuda(), beta.cuda(), rm.clone().cuda(), rv.clone().cuda(), eps, momentum)
    return spec, gamma, beta, rm, rv


@pytest.mark.parametrize("groups,C,act,nres,window", [(1, 64, 1, 0, False), (2, 96, 1, 1, True), (2, 32, 0, 2, True),
                                                     (1, 300, 1, 1, False), (4, 16, 1, 0, False)])
def test_batchnorm_train_forward_backward_match_torch(groups, C, act, nres, window):
    from thinktwice_amd import autodiff, ops
    g = torch.Generator().manual_seed(groups * 100 + C)
    N, H, W = 2 * groups, 5, 7
    z = torch.randn(N, H, W, C, generator=g) * 2.0 + torch.randn(C, generator=g)
    spec, gamma, beta, rm, rv = _spec(C, g, eps=1e-3, momentum=0.01)
    Ct, coff = (C + 40, 24) if window else (C, 0)
    res = [torch.randn(N, H, W, Ct, generator=g) for _ in range(nres)]
    R = torch.randn(N, H, W, C, generator=g)                      # d(loss)/d(out)
    # ---- torch reference: one F.batch_norm per group, in group order (running statistics see group 0 first)
    zt = z.clone().requires_grad_(True)
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rt = [r.clone().requires_grad_(True) for r in res]
    rmt, rvt = rm.clone(), rv.clone()
    per = N // groups
    ys = []
    for k in range(groups):
        zz = zt[k * per:(k + 1) * per].permute(0, 3, 1, 2)
        ys.append(F.batch_norm(zz, rmt, rvt, gt, bt, True, 0.01, 1e-3).permute(0, 2, 3, 1))
    y = torch.cat(ys, 0)
    for r in rt:
        y = y + r[..., coff:coff + C]
    if act == 1:
        y = torch.relu(y)
    (y * R).sum().backward()
    # ---- HIP
    zd = z.cuda()
    out = torch.full((N, H, W, Ct), 7.0, device="cuda")
    rd = [r.cuda() for r in res]
    with autodiff.Tape() as tape:
        ops.batchnorm_train(zd, spec, act, res1=rd[0] if nres > 0 else None, res1_coff=coff,
                            res2=rd[1] if nres > 1 else None, res2_coff=coff, out=out, out_coff=coff, groups=groups)
        tape.seed(out[..., coff:coff + C], R)
        tape.backward()
        dz = tape.grad(zd).cpu()
        dres = [tape.grad(r).cpu() for r in rd]
    torch.cuda.synchronize()
    got = out.cpu()
    assert float((got[..., coff:coff + C] - y.detach()).abs().max()) < 2e-5 * float(y.detach().abs().max())
    if window:
        assert bool((got[..., :coff] == 7.0).all()) and bool((got[..., coff + C:] == 7.0).all())
    assert float((spec.running_mean.cpu() - rmt).abs().max()) < 1e-6
    assert float((spec.running_var.cpu() - rvt).abs().max()) < 1e-5
    assert float((dz - zt.grad).abs().max()) < 2e-4 * float(zt.grad.abs().max())
    for d, r in zip(dres, rt):
        assert float((d[..., coff:coff + C] - r.grad[..., coff:coff + C]).abs().max()) < 1e-5
    dg, db = tape.param_grads["layer.bn.weight"].cpu(), tape.param_grads["layer.bn.bias"].cpu()
    assert float((dg - gt.grad).abs().max()) < 2e-4 * float(gt.grad.abs().max())
    assert float((db - bt.grad).abs().max()) < 2e-4 * float(bt.grad.abs().max())


def test_batchnorm_train_sparse_rows_use_the_device_row_count():
    from thinktwice_amd import autodiff, ops
    g = torch.Generator().manual_seed(4)
    M, live, C = 5000, 4321, 48
    z = torch.randn(M, C, generator=g)
    z[live:] = 1e6                                   # garbage beyond the live rows must not enter the statistics
    spec, gamma, beta, rm, rv = _spec(C, g)
    res = torch.randn(M, C, generator=g)
    R = torch.randn(live, C, generator=g)
    zt, rt = z[:live].clone().requires_grad_(True), res[:live].clone().requires_grad_(True)
    y = torch.relu(F.batch_norm(zt, None, None, gamma, beta, True, 0.0, 1e-5) + rt)
    (y * R).sum().backward()
    zd, rd = z.cuda(), res.cuda()
    m_dev = torch.tensor([live], dtype=torch.int32, device="cuda")
    with autodiff.Tape() as tape:
        out = ops.batchnorm_train(zd, spec, 1, res1=rd, m_dev=m_dev)
        tape.seed(out[:live], R)
        tape.backward()
        dz = tape.grad(zd).cpu()
    torch.cuda.synchronize()
    assert float((out[:live].cpu() - y.detach()).abs().max()) < 2e-5
    assert float((dz[:live] - zt.grad).abs().max()) < 2e-4 * float(zt.grad.abs().max())
    assert float(dz[live:].abs().max()) == 0.0


def test_dropout_replays_a_mask_and_its_own_generator_keeps_about_half():
    from thinktwice_amd import autodiff, ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 8, 8, 32, generator=g)
    mask = (torch.rand(x.shape, generator=g) < 0.5).to(torch.uint8)
    R = torch.randn(x.shape, generator=g)
    ops.DROPOUT_MASKS = iter([mask])
    try:
        xd = x.cuda()
        with autodiff.Tape() as tape:
            y = ops.dropout(xd, 0.5)
            tape.seed(y, R)
            tape.backward()
            dx = tape.grad(xd).cpu()
    finally:
        ops.DROPOUT_MASKS = None
    assert torch.equal(y.cpu(), x * mask * 2.0) and torch.equal(dx, R * mask * 2.0)
    y2 = ops.dropout(xd, 0.5).cpu()
    kept = float((y2 != 0).float().mean())
    assert 0.45 < kept < 0.55 and torch.equal(y2[y2 != 0], (x * 2.0)[y2 != 0])
    y3 = ops.dropout(xd, 0.5).cpu()
    assert not torch.equal(y2 != 0, y3 != 0)           # a new mask per call


def test_row_batchnorm_running_statistics_take_one_update_per_sweep():
    """the reference runs DepthNet's BatchNorm1d(22) once per SWEEP on the key frame's camera vector (lss.py:689-714),
    so its running statistics take T momentum updates per iteration with the same batch statistics.  `running_updates=T` must
    leave exactly what torch leaves after T train-mode calls on the same input, and the same normalised output."""
    from thinktwice_amd import layers
    g = torch.Generator().manual_seed(22)
    C, R, T = 22, 16, 2
    x = torch.randn(R, C, generator=g) * torch.rand(C, generator=g) * 30 + torch.randn(C, generator=g) * 5
    sd = {"bn.weight": torch.rand(C, generator=g) + 0.5, "bn.bias": torch.randn(C, generator=g) * 0.2,
          "bn.running_mean": torch.randn(C, generator=g), "bn.running_var": torch.rand(C, generator=g) + 0.5}
    ref = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        ref.weight.copy_(sd["bn.weight"]); ref.bias.copy_(sd["bn.bias"])
        ref.running_mean.copy_(sd["bn.running_mean"]); ref.running_var.copy_(sd["bn.running_var"])
    ref.train()
    for _ in range(T):
        y_ref = ref(x)
    bn = layers.bn_affine(sd, "bn", "cuda")
    old = layers.BN_TRAIN
    layers.BN_TRAIN = True
    try:
        y = bn(x.cuda(), running_updates=T)
    finally:
        layers.BN_TRAIN = old
    assert float((y.cpu()[:, :C] - y_ref.detach()).abs().max()) < 2e-5
    assert float((bn.spec.running_mean.cpu() - ref.running_mean).abs().max()) < 1e-5
    assert float(((bn.spec.running_var.cpu() - ref.running_var) / ref.running_var).abs().max()) < 1e-5
