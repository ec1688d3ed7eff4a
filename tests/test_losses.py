"""Loss reductions (csrc/losses.hip, row A24) on their own: each tt_loss_* entry against the oracle's restatement of the
reference formulas (oracle/train_ref.py: thinktwice_decoder.py:536-637, encoder_decoder_framework.py:172-190, 441-489,
utils.py:31-47) on random inputs, including the edge cases the reference code has branches for."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _red():
    from thinktwice_amd.losses import LossReducer
    return LossReducer("cuda")


def test_beta_kl_and_action_offsets_match_torch_distributions():
    from torch.distributions import Beta, kl_divergence
    from oracle import train_ref as TR
    g = torch.Generator().manual_seed(3)
    red = _red()
    for N, R, inner_shape in ((2, 5, (2,)), (8, 5, (3, 2)), (3, 1, (2,))):
        # Beta parameters on both sides of 1 (all four branches of _get_action_beta), small and large
        t_a = torch.rand(N, *inner_shape, generator=g) * 3.0 + 0.05
        t_b = torch.rand(N, *inner_shape, generator=g) * 3.0 + 0.05
        p_a = torch.rand(N, R, *inner_shape, generator=g) * 6.0 + 0.02
        p_b = torch.rand(N, R, *inner_shape, generator=g) * 6.0 + 0.02
        want = kl_divergence(Beta(t_a.unsqueeze(1), t_b.unsqueeze(1)), Beta(p_a, p_b)).mean() * 15.0
        got = red.beta_kl(t_a, t_b, p_a.cuda(), p_b.cuda(), 15.0)
        assert abs(float(got) - float(want)) < 2e-5 * abs(float(want)) + 1e-6, (float(got), float(want))
    a = torch.tensor([[0.5, 2.0], [2.0, 0.5], [0.4, 0.3], [3.0, 2.5], [1.0, 1.0]])
    b = torch.tensor([[2.0, 0.7], [0.5, 3.0], [0.2, 0.9], [1.5, 4.0], [1.0, 2.0]])
    ta, tb = a.flip(0).contiguous(), b.flip(0).contiguous()
    want = F.l1_loss(TR.action_beta(a, b), TR.action_beta(ta, tb), reduction="none").mean(0)
    got = red.l1_cols(a.cuda(), ta, b.cuda(), tb).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6, atol=1e-7)


def test_smooth_l1_broadcast_clamp_and_elementwise():
    g = torch.Generator().manual_seed(5)
    red = _red()
    x = torch.randn(3, 5, 4, 7, generator=g) * 4.0          # |d| on both sides of 1 and of the clamp at 5
    t = torch.randn(3, 4, 7, generator=g) * 4.0
    ref = torch.clamp(F.smooth_l1_loss(x, t.unsqueeze(1).expand_as(x), reduction="none"), min=-5.0, max=5.0).mean() * 0.25
    assert abs(float(red.smooth_l1(x.cuda(), t, clamp_max=5.0, scale=0.25)) - float(ref)) < 1e-6 * float(ref) + 1e-7
    ref0 = F.smooth_l1_loss(x, torch.zeros_like(x))
    assert abs(float(red.smooth_l1(x.cuda())) - float(ref0)) < 1e-6 * float(ref0)
    v, gt = torch.randn(6, 1, generator=g), torch.randn(6, 1, generator=g)
    none = red.smooth_l1(v.cuda(), gt, scale=0.001, reduce=False).cpu()
    np.testing.assert_allclose(none.numpy(), (F.smooth_l1_loss(v, gt, reduction="none") * 0.001).numpy(), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("BN,H,W", [(2, 32, 64), (8, 64, 128)])
def test_seg_focal_and_depth_bce_match_oracle(BN, H, W):
    from oracle import train_ref as TR
    g = torch.Generator().manual_seed(BN + H)
    red = _red()
    B, N = 1, BN
    # segmentation: 12 classes in a 16-wide channel-last row, some labels = 255 (ignored)
    seg = torch.randn(BN, 12, H // 2, W // 2, generator=g) * 2.0
    lab = torch.randint(0, 12, (B, N, H, W), generator=g).float()
    lab[torch.rand(B, N, H, W, generator=g) < 0.1] = 255.0
    want = TR.seg_loss(seg, lab)
    seg_cl = torch.zeros(BN, H // 2, W // 2, 16)
    seg_cl[..., :12] = seg.permute(0, 2, 3, 1)
    seg_cl[..., 12:] = 50.0                                    # padding channels must not enter the softmax
    got = red.seg_focal(seg_cl.cuda().contiguous(), lab, num_classes=12, factor=2)
    assert abs(float(got) - float(want)) < 2e-5 * abs(float(want)), (float(got), float(want))
    # depth: sparse metres with out-of-range values on both sides, factor 16
    d_bound = [1.0, 41.0, 0.5]
    D = 80
    dep = torch.rand(B, N, H, W, generator=g) * 50.0
    dep[torch.rand(B, N, H, W, generator=g) < 0.97] = 0.0
    logits = torch.randn(BN, D, H // 16, W // 16, generator=g) * 3.0
    want = TR.depth_loss(logits, dep, d_bound, 16)
    got = red.depth_bce(logits.permute(0, 2, 3, 1).contiguous().cuda(), dep, d_bound, 16)
    assert abs(float(got) - float(want)) < 2e-5 * abs(float(want)), (float(got), float(want))
    # no foreground cell at all: sum / max(1, 0) = 0
    got0 = red.depth_bce(logits.permute(0, 2, 3, 1).contiguous().cuda(), torch.zeros_like(dep), d_bound, 16)
    assert float(got0) == 0.0 and float(TR.depth_loss(logits, torch.zeros_like(dep), d_bound, 16)) == 0.0


def test_smooth_l1_and_beta_kl_gradients_match_autograd():
    """The gradient half of the two decoder loss forms (tt_loss_smooth_l1_bwd, tt_loss_beta_kl_bwd) as the training tape
    records them: d[mean-reduced term]/d(prediction) with a broadcast target, the clamp (zero gradient where it is active),
    the unreduced (B, 1) form that `_parse_losses` averages, and predictions that are strided VIEWS of a larger tensor
    (mu / sigma columns of the stacked control tensor)."""
    from torch.distributions import Beta, kl_divergence
    from thinktwice_amd import autodiff
    g = torch.Generator().manual_seed(11)
    red = _red()
    # smooth-L1: broadcast over the refinement dimension, residuals on both sides of 1 and of the clamp
    x = (torch.randn(4, 5, 3, 7, generator=g) * 4.0).requires_grad_(True)
    t = torch.randn(4, 3, 7, generator=g)
    want = (torch.clamp(F.smooth_l1_loss(x, t.unsqueeze(1).expand_as(x), reduction="none"), max=5.0).mean() * 0.25)
    want.backward()
    xd = x.detach().cuda()
    with autodiff.Tape() as tape:
        got = red.smooth_l1(xd, t, clamp_max=5.0, scale=0.25)
        tape.backward()
    assert abs(float(got) - float(want)) < 1e-6 * abs(float(want))
    assert float((tape.grad(xd).cpu() - x.grad).abs().max()) < 1e-7
    assert int((x.grad == 0).sum()) > 0                       # the clamp was active somewhere
    # unreduced form: (B, 1) values, averaged by parse_losses
    v = torch.randn(6, 1, generator=g, requires_grad=True)
    tv = torch.randn(6, 1, generator=g)
    (F.smooth_l1_loss(v, tv, reduction="none") * 0.001).mean().backward()
    vd = v.detach().cuda()
    with autodiff.Tape() as tape:
        red.smooth_l1(vd, tv, scale=0.001, reduce=False)
        tape.backward()
    assert float((tape.grad(vd).cpu() - v.grad).abs().max()) < 1e-9
    # Beta KL on strided views: ct (B, R, 4, 4) -> mu = ct[:, :, 0, :2], sigma = ct[:, :, 0, 2:]
    ct = (torch.rand(3, 6, 4, 4, generator=g) * 4.0 + 0.05).requires_grad_(True)
    ta = torch.rand(3, 2, generator=g) * 3.0 + 0.1
    tb = torch.rand(3, 2, generator=g) * 3.0 + 0.1
    (kl_divergence(Beta(ta.unsqueeze(1), tb.unsqueeze(1)), Beta(ct[:, :, 0, :2], ct[:, :, 0, 2:])).mean() * 15.0).backward()
    cd = ct.detach().cuda()
    with autodiff.Tape() as tape:
        red.beta_kl(ta, tb, cd[:, :, 0, :2], cd[:, :, 0, 2:], 15.0)
        tape.backward()
    err = float((tape.grad(cd).cpu() - ct.grad).abs().max() / ct.grad.abs().max())
    assert err < 1e-5, err
    assert float(tape.grad(cd)[:, :, 1:].abs().max()) == 0.0  # nothing outside the two views
