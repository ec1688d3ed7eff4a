"""Kernels of the composite decoder (csrc/dec_spatial.hip, bf16x3 arithmetic) against the oracle's functions of the
same reference code: SpatialGRU (dense_heads/utils.py:53-106), grid2feat (encoder_decoder_framework.py:228-234) and the
BEV update (thinktwice_decoder.py:221-225,257).  Tolerance 1e-3 of each tensor's max (measured ~1e-5..1e-4); the
end-to-end composite decoder is covered by tests/test_decoder.py (ids "composite") and tests/test_forward.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


@pytest.fixture(scope="module")
def sd_cfg():
    from thinktwice_amd import config, params
    cfg = config.model_config(final_dim=(128, 256))
    return params.init_params(cfg, seed=0, parts=("fusion", "decoder")), cfg


@pytest.mark.parametrize("B", [1, 3])
def test_gru_kernel_matches_oracle(sd_cfg, B):
    from oracle import model_ref as M
    from thinktwice_amd import decoder_fused as DF, ops
    sd, _ = sd_cfg
    p = "decoder.decoder_layers.1.prediction_module.spatial_gru"
    g = torch.Generator().manual_seed(B)
    inp6 = torch.randn(B, 4, 6, generator=g)
    state = torch.randn(B, 32, 21, 21, generator=g) * 0.5
    with torch.no_grad():
        ref = M.spatial_gru(sd, p, inp6[..., None, None].expand(B, 4, 6, 21, 21), state)       # (B,4,32,21,21)
    w = DF.prep_gru(sd, p, "cuda")
    fut = torch.full((B, 4, 441, 32), float("nan"), device="cuda")
    ops.dec_gru(w, inp6.cuda(), state.permute(0, 2, 3, 1).reshape(B, 441, 32).contiguous().cuda(), fut)
    torch.cuda.synchronize()
    got = fut.cpu().view(B, 4, 21, 21, 32).permute(0, 1, 4, 2, 3)
    assert _rel(got, ref) < 1e-3, _rel(got, ref)


def test_gru_handover_timeout_is_loud_and_recoverable(sd_cfg):
    """tt_dec_gru runs two workgroups per sample that hand the state / the update gate over through flags.  With the wait bound
    forced to 0 polls (tt_mlp_chain_wide_set_max_spin, shared with the wide chain) the first wait gives up: the future maps are
    NaN, the device's fault word is set and the next launch is refused; after clear_device_faults() the results are the
    earlier ones again, bit for bit."""
    from thinktwice_amd import _lib, decoder_fused as DF, ops
    sd, _ = sd_cfg
    L = _lib.lib()
    w = DF.prep_gru(sd, "decoder.decoder_layers.0.prediction_module.spatial_gru", "cuda")
    g = torch.Generator().manual_seed(9)
    B = 2
    inp6 = torch.randn(B, 4, 6, generator=g).cuda()
    state = (torch.randn(B, 441, 32, generator=g) * 0.5).cuda()

    def run():
        fut = torch.zeros(B, 4, 441, 32, device="cuda")
        ops.dec_gru(w, inp6, state, fut)
        torch.cuda.synchronize()
        return fut.cpu()

    assert ops.device_faults() == 0
    good = run()
    assert torch.isfinite(good).all()
    try:
        L.tt_mlp_chain_wide_set_max_spin(0)
        bad = run()
        assert torch.isnan(bad).any(), "a timed-out hand-over must poison the outputs"
        assert ops.device_faults() == 1
        with pytest.raises(_lib.TTError, match="gave up waiting"):
            run()                                            # sticky: the next launch is refused
    finally:
        L.tt_mlp_chain_wide_set_max_spin(-1)
        ops.clear_device_faults()
    for _ in range(3):
        assert torch.equal(run(), good)


@pytest.mark.parametrize("N", [6, 35])      # 35 maps: more than one group of maps at every level of the column-split tail
def test_flatten_kernel_matches_oracle(sd_cfg, N):
    """tt_dec_flatten (one workgroup per map up to conv10_4, then dec_tail_conv_kernel per layer over all maps' rows)."""
    from oracle import model_ref as M
    from thinktwice_amd import decoder_fused as DF, ops
    sd, _ = sd_cfg
    g = torch.Generator().manual_seed(5)
    f21 = torch.randn(N, 32, 21, 21, generator=g).abs() * 0.7
    with torch.no_grad():
        flat_r, (f10, f4, f2) = M.flatten_tail(sd, f21)
    w = DF.prep_flatten(sd, "cuda")
    flat, mids = ops.dec_flatten(w, f21.permute(0, 2, 3, 1).reshape(N, 441, 32).contiguous().cuda(), want_mids=True)
    torch.cuda.synchronize()
    mids = mids.cpu()
    got10 = mids[:, :6400].view(N, 10, 10, 64).permute(0, 3, 1, 2)
    got4 = mids[:, 6400:6400 + 2048].view(N, 4, 4, 128).permute(0, 3, 1, 2)
    got2 = mids[:, 8448:].view(N, 2, 2, 256).permute(0, 3, 1, 2)
    errs = {"f10": _rel(got10, f10), "f4": _rel(got4, f4), "f2": _rel(got2, f2), "flat": _rel(flat.cpu(), flat_r)}
    assert max(errs.values()) < 1e-3, errs


@pytest.mark.parametrize("B", [1, 8])
def test_bev_update_kernel_matches_oracle(sd_cfg, B):
    from oracle import model_ref as M
    from thinktwice_amd import decoder_fused as DF, ops
    sd, _ = sd_cfg
    q = "decoder.decoder_layers.2"
    g = torch.Generator().manual_seed(11 + B)
    bev = torch.randn(B, 32, 21, 21, generator=g) * 0.5
    hb = torch.randn(B, 2048, generator=g).abs() * 0.3
    with torch.no_grad():
        x = torch.cat([bev, hb[..., None, None].expand(B, 2048, 21, 21)], 1)
        ref = M.conv(sd, q + ".BEV_feat_update_module.2", F.relu(M.conv(sd, q + ".BEV_feat_update_module.0", x, 1, 1)), 1, 1) + bev
    w = DF.prep_bev_update(sd, q, "cuda")
    G = torch.empty(B, 1152, device="cuda")
    ops.mlp_chain(hb.cuda(), [{"lin": w["G"], "src": -1, "out": (G, 0)}], n_split=5)
    # the broadcast-channel term itself: G[b, tap*128 + n] = W0[n, 32:, tap] . hb[b]
    W0 = sd[q + ".BEV_feat_update_module.0.weight"]
    g_ref = torch.einsum("nckl,bc->bkln", W0[:, 32:], hb).reshape(B, 1152)
    assert _rel(G.cpu(), g_ref) < 1e-4
    out = torch.full((B, 441, 32), float("nan"), device="cuda")
    ops.dec_bev_update(w, bev.permute(0, 2, 3, 1).reshape(B, 441, 32).contiguous().cuda(), G, out)
    torch.cuda.synchronize()
    got = out.cpu().view(B, 21, 21, 32).permute(0, 3, 1, 2)
    assert _rel(got, ref) < 1e-3, _rel(got, ref)


def test_msda_sample_ln_skips_only_the_slots_no_one_reads():
    """tt_msda_sample_ln with the device `max_len` of tt_look_project_pack: rows of slots < max_len are bit-identical to the
    unskipped launch, rows of slots >= max_len are left untouched (tt_sca_reduce_ln sums k < max_len only, so the look feature
    is unchanged)."""
    from thinktwice_amd import ops
    B = 2
    g = torch.Generator().manual_seed(5)
    level_hw = [(16, 32), (8, 16), (4, 8), (2, 4)]
    S = sum(h * w for h, w in level_hw)
    value = torch.randn(B * 4, S, 512, generator=g).cuda()                 # two layers' value maps side by side
    R = B * 4 * 120
    off = (torch.randn(R, 512, generator=g) * 2).cuda()
    aw = torch.randn(R, 256, generator=g).cuda()
    ref = torch.rand(B, 4, 120, 2, generator=g).cuda()
    gamma, beta = torch.randn(256, generator=g).cuda(), torch.randn(256, generator=g).cuda()
    full, full_ln = ops.msda_sample_ln(value, off, aw, ref, level_hw, B, 256, gamma, beta)
    ml = torch.tensor([37], dtype=torch.int32, device="cuda")
    import thinktwice_amd.ops as O
    real_empty = torch.empty
    try:                                                                    # the skipped rows keep whatever the buffers held: NaN
        torch.empty = lambda *a, **k: real_empty(*a, **k).fill_(float("nan")) if k.get("dtype") == torch.float32 else real_empty(*a, **k)
        part, part_ln = O.msda_sample_ln(value, off, aw, ref, level_hw, B, 256, gamma, beta, max_len=ml)
    finally:
        torch.empty = real_empty
    torch.cuda.synchronize()
    slot = torch.arange(R, device="cuda") % 120
    live = slot < 37
    assert torch.equal(part[live], full[live]) and torch.equal(part_ln[live], full_ln[live])
    assert torch.isnan(part[~live]).all() and torch.isnan(part_ln[~live]).all()
    y = torch.randn(R, 256, generator=g).cuda()
    a = ops.sca_reduce_ln(torch.where(live[:, None], y, torch.full_like(y, float("nan"))), ml, B, torch.ones(1024).cuda(),
                          torch.zeros(1024).cuda())
    b = ops.sca_reduce_ln(y, ml, B, torch.ones(1024).cuda(), torch.zeros(1024).cuda())
    assert torch.equal(a, b)                                                # the consumer never touches the skipped rows


def test_sample_first_project_after_equals_projecting_every_position():
    """tt_msda_sample_proj_ln (round 6): the attention rows from the RAW fpn_linear maps -- weighted sum of the sampled 256-channel
    rows per head, then the head's slice of value_proj, bias and (level, camera) embedding shift weighted by the in-bounds corner
    weight -- against the reference's order (multi_scale_deformable_attn_function.py:474: value_proj of every position, then
    sample): tt_conv2d_fwd (exact f32) + tt_msda_sample_ln.  Offsets large enough that many corners fall outside (zero padding,
    where the bias must NOT contribute); exact-f32 arithmetic on both sides, different summation order: 2e-5 of the max."""
    from thinktwice_amd import ops
    B = 2
    g = torch.Generator().manual_seed(7)
    level_hw = [(16, 32), (8, 16), (4, 8), (2, 4)]
    maps = [torch.randn(B * 4, h, w, 256, generator=g).cuda() for h, w in level_hw]
    W = (torch.randn(256, 256, generator=g) * 256 ** -0.5).cuda()
    bias = torch.randn(256, generator=g).cuda()
    vshift = torch.randn(4, 4, 256, generator=g).cuda()                     # (level, camera, channel)
    R = B * 4 * 120
    off = (torch.randn(R, 512, generator=g) * 3).cuda()
    aw = torch.randn(R, 256, generator=g).cuda()
    ref = torch.rand(B, 4, 120, 2, generator=g).cuda()
    gamma, beta = torch.randn(256, generator=g).cuda(), torch.randn(256, generator=g).cuda()
    # reference order: project every position (+ bias + per-(level, camera) shift), then sample
    S = sum(h * w for h, w in level_hw)
    value = torch.empty(B * 4, S, 256, device="cuda")
    start = 0
    for l, (m, (h, w)) in enumerate(zip(maps, level_hw)):
        ops.conv2d(m, W.view(256, 1, 1, 256).contiguous(), shift=bias, shift_n=vshift[l].contiguous(), shift_n_mod=4,
                   out=value[:, start:start + h * w].unflatten(1, (h, w)), out_nstride=S * 256)
        start += h * w
    want, want_ln = ops.msda_sample_ln(value, off, aw, ref, level_hw, B, 0, gamma, beta)
    got, got_ln = ops.msda_sample_proj_ln(maps, off, aw, ref, B, W.t().contiguous(), bias, vshift.contiguous(), gamma, beta)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < 2e-5 * scale, float((got - want).abs().max()) / scale
    assert float((got_ln - want_ln).abs().max()) < 1e-3 * float(want_ln.abs().max())      # (LayerNorm amplifies by 1 / std of a row)
    ml = torch.tensor([29], dtype=torch.int32, device="cuda")
    part, _ = ops.msda_sample_proj_ln(maps, off, aw, ref, B, W.t().contiguous(), bias, vshift.contiguous(), gamma, beta, max_len=ml)
    live = (torch.arange(R, device="cuda") % 120) < 29
    assert torch.equal(part[live], got[live])
