"""tt_mlp_chain (the decoder's row-batched MLP chains in one launch, bf16x3 arithmetic) vs the same chain of
nn.Linear layers in torch f32 on the CPU.  Tolerance 1e-4 of the output's max per stage output."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACT = {0: lambda t: t, 1: F.relu, 2: torch.sigmoid, 3: F.gelu}


def _lin(g, n, k, scale=None):
    return torch.randn(n, k, generator=g) * (scale or k ** -0.5), torch.randn(n, generator=g) * 0.1


@pytest.mark.parametrize("R", [5, 32, 33, 3840])
def test_look_query_chain_matches_torch(R):
    """query_linear.1 (1544 -> 512, GELU) -> .3 (512 -> 256, GELU) -> {sampling_offsets 256 -> 512,
    attention_weights 256 -> 256}: two outputs fanned out of one LDS intermediate, zero-padded input rows."""
    from thinktwice_amd import ops
    g = torch.Generator().manual_seed(R)
    x = torch.zeros(R, 1552)
    x[:, :1543] = torch.randn(R, 1543, generator=g)
    w1, b1 = _lin(g, 512, 1544)
    w3, b3 = _lin(g, 256, 512)
    wo, bo = _lin(g, 512, 256)
    wa, ba = _lin(g, 256, 256)
    q = F.gelu(F.linear(F.gelu(F.linear(x[:, :1544], w1, b1)), w3, b3))
    ref_off, ref_aw = F.linear(q, wo, bo), F.linear(q, wa, ba)
    dev = "cuda"
    L1, L3 = ops.ChainLinear(w1, b1, act=3), ops.ChainLinear(w3, b3, act=3)
    LO, LA = ops.ChainLinear(wo, bo), ops.ChainLinear(wa, ba)
    off = torch.full((R, 512), float("nan"), device=dev)
    aw = torch.full((R, 300), float("nan"), device=dev)
    ops.mlp_chain(x.to(dev), [{"lin": L1, "src": -1}, {"lin": L3, "src": 0},
                              {"lin": LO, "src": 1, "out": (off, 0)}, {"lin": LA, "src": 1, "out": (aw, 44)}])
    torch.cuda.synchronize()
    for got, ref in ((off.cpu(), ref_off), (aw.cpu()[:, 44:], ref_aw)):
        assert float((got - ref).abs().max() / ref.abs().max()) < 1e-4
    assert torch.isnan(aw[:, :44]).all()


def test_residual_side_input_and_narrow_heads():
    """ffn-style residual (out = W2 gelu(W1 x) + x), a side input of 2 leading columns (cat([wp, h]) @ W^T), a 2-wide
    head (traj offset) and a 4-wide head, R not a multiple of 32."""
    from thinktwice_amd import ops
    g = torch.Generator().manual_seed(7)
    R = 45
    x = torch.randn(R, 256, generator=g)
    wp = torch.randn(R, 2, generator=g)
    w1, b1 = _lin(g, 512, 256)              # (a 1024-wide hidden + a kept 256-wide output would exceed 160 KiB of LDS)
    w2, b2 = _lin(g, 256, 512)
    wt, bt = _lin(g, 256, 258)              # [wp (2) | y (256)] -> 256
    wh, bh = _lin(g, 2, 256)
    wc, bc = _lin(g, 4, 256)
    y = F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2) + x
    t = F.relu(F.linear(torch.cat([wp, y], 1), wt, bt))
    ref_h, ref_c = F.linear(t, wh, bh), F.softplus(F.linear(t, wc, bc))
    dev = "cuda"
    xd, wpd = x.to(dev), wp.to(dev)
    yo = torch.empty(R, 256, device=dev)
    ho = torch.empty(R, 2, device=dev)
    co = torch.empty(R, 4, device=dev)
    stages = [{"lin": ops.ChainLinear(w1, b1, act=3), "src": -1},
              {"lin": ops.ChainLinear(w2, b2), "src": 0, "res": (xd, 0), "out": (yo, 0)},
              {"lin": ops.ChainLinear(wt, bt, act=1, side_k=2), "src": 1, "side": wpd},
              {"lin": ops.ChainLinear(wh, bh), "src": 2, "out": (ho, 0)},
              {"lin": ops.ChainLinear(wc, bc, act=4), "src": 2, "out": (co, 0)}]
    ops.mlp_chain(xd, stages)
    torch.cuda.synchronize()
    for got, ref in ((yo.cpu(), y), (ho.cpu(), ref_h), (co.cpu(), ref_c)):
        assert float((got - ref).abs().max() / ref.abs().max()) < 1e-4, float((got - ref).abs().max() / ref.abs().max())


def test_wide_hidden_chain_and_lds_budget():
    """ffn shape 256 -> 1024 -> 256 (+ residual): the 1024-wide intermediate is 131.6 KB of LDS per 32 rows -- fits
    alone; asking to ALSO keep its 256-wide successor for a third stage must be refused, not overflow."""
    from thinktwice_amd import _lib, ops
    g = torch.Generator().manual_seed(9)
    R = 70
    x = torch.randn(R, 256, generator=g)
    w1, b1 = _lin(g, 1024, 256)
    w2, b2 = _lin(g, 256, 1024)
    ref = F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2) + x
    xd = x.cuda()
    y = torch.empty(R, 256, device="cuda")
    L1, L2 = ops.ChainLinear(w1, b1, act=3), ops.ChainLinear(w2, b2)
    ops.mlp_chain(xd, [{"lin": L1, "src": -1}, {"lin": L2, "src": 0, "res": (xd, 0), "out": (y, 0)}])
    torch.cuda.synchronize()
    assert float((y.cpu() - ref).abs().max() / ref.abs().max()) < 1e-4
    w3, b3 = _lin(g, 8, 256)
    with pytest.raises(_lib.TTError):
        ops.mlp_chain(xd, [{"lin": L1, "src": -1}, {"lin": L2, "src": 0}, {"lin": ops.ChainLinear(w3, b3), "src": 1,
                                                                          "out": (torch.empty(R, 8, device="cuda"), 0)}])


def test_chain_rejects_bad_wiring():
    from thinktwice_amd import _lib, ops
    g = torch.Generator().manual_seed(1)
    w1, b1 = _lin(g, 64, 32)
    w2, b2 = _lin(g, 16, 48)                # K = 48 does not match the 64 outputs of stage 0
    x = torch.zeros(8, 32, device="cuda")
    with pytest.raises(_lib.TTError):
        ops.mlp_chain(x, [{"lin": ops.ChainLinear(w1, b1), "src": -1}, {"lin": ops.ChainLinear(w2, b2), "src": 0}])
