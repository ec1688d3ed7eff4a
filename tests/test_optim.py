"""Optimizer half of the training step: HIP global-norm clip + fused AdamW over a flat buffer vs
torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW on the CPU (what the reference's OptimizerHook runs,
configs/thinktwice.py:282-287), three steps, one of them clipped."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_flat_adamw_matches_torch_adamw_with_grad_clip():
    from thinktwice_amd.optim import FlatAdamW
    n = 1_000_003
    g = torch.Generator().manual_seed(11)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * s for s in (1.0, 0.01, 0.3)]       # norms ~1000, ~10, ~300 vs max 100
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7)
    dev_p = p0.clone().cuda()
    dev_g = torch.zeros(n, device="cuda")
    mine = FlatAdamW(dev_p, dev_g, lr=1e-4, weight_decay=1e-7, max_grad_norm=100.0)
    for i, gr in enumerate(grads):
        ref_p.grad = gr.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_([ref_p], 100.0)
        opt.step()
        dev_g.copy_(gr)
        ns = mine.step().cpu()
        assert abs(float(ns[0]) - float(ref_norm)) < 1e-4 * float(ref_norm)
        # torch's f32 vector_norm is itself ~1e-5 off the exact norm at n = 1e6 (300.0512 vs 300.0546 on step 3), so the
        # clip factor can only be compared at that level (an absolute 1e-6 bound was what failed on the hardware run)
        want = min(1.0, 100.0 / (float(ref_norm) + 1e-6))
        assert abs(float(ns[1]) - want) < 1e-4 * want
        err = float((dev_p.cpu() - ref_p.detach()).abs().max())
        assert err < 2e-6, (i, err)            # a few f32 ulps at |p| ~ 4 (different op fusion than torch)
    # the update actually moved the parameters by about lr per step
    assert 1e-4 < float((dev_p.cpu() - p0).abs().max()) < 1e-3
    # and the moments match torch's state
    st = opt.state[ref_p]
    assert float((mine.m.cpu() - st["exp_avg"]).abs().max()) < 2e-6
    # the C ABI takes the betas as f32: (1 - 0.999f) differs from torch's double-computed 1 - 0.999 by 4.7e-5 relative
    assert float((mine.v.cpu() - st["exp_avg_sq"]).abs().max()) < 1e-4 * float(st["exp_avg_sq"].max())
