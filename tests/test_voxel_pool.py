"""Voxel pooling: oracle vs golden on CPU; HIP vs oracle/golden on GPU."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import c_ref
from oracle import lss_geometry as og

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ----------------------------------------------------------------------------- CPU: oracle pins
def test_oracle_voxel_pool_matches_reference_wrapper_golden(golden_dir):
    f = _load(golden_dir, "f3_voxel_pool.npz")
    out, memo = c_ref.voxel_pool_fwd(f["geom"], f["feats"], f["voxel_num"])
    np.testing.assert_allclose(out.transpose(0, 3, 1, 2), f["out"], rtol=1e-5, atol=1e-5)
    gi = c_ref.voxel_pool_bwd(memo, f["grad_out"].transpose(0, 2, 3, 1))
    np.testing.assert_array_equal(gi, f["grad_in"])
    # .int() truncation toward zero (lss.py:630-631): (-1,0) -> 0
    np.testing.assert_array_equal(torch.from_numpy(f["trunc_float"]).int().numpy(), f["trunc_int"])
    assert f["trunc_int"][0] == 0 and f["trunc_int"][2] == 0 and f["trunc_int"][-1] == -1


def test_oracle_geometry_matches_reference_golden(golden_dir):
    f = _load(golden_dir, "f12_geometry.npz")
    fr = og.create_frustum()
    assert tuple(fr.shape) == tuple(f["frustum_shape"])
    np.testing.assert_array_equal(fr[[0, 0, -1, -1], [0, -1, 0, -1], [0, -1, -1, 0]].numpy(),
                                  f["frustum_corners"])
    vs, vc, vn = og.voxel_constants()
    np.testing.assert_array_equal(vs.numpy(), f["voxel_size"])
    np.testing.assert_array_equal(vc.numpy(), f["voxel_coord"])
    np.testing.assert_array_equal(vn.numpy(), f["voxel_num"])
    from thinktwice_amd import synth
    metas = synth.make_img_metas(2, curr2key=f["curr2key"])
    intr, ida, s2e, l2i, cur_ida = og.assemble_camera_mats(metas)
    for a, k in ((intr, "intrin"), (ida, "ida"), (s2e, "sensor2ego"), (l2i, "lidar2img"),
                 (cur_ida, "cur_ida")):
        np.testing.assert_array_equal(a.numpy(), f[k])
    geom = og.get_geometry(fr, s2e[:, -1], intr[:, -1], ida[:, -1])
    idx = og.voxel_index(geom, vc, vs)
    sel = torch.from_numpy(f["sample_index"])
    np.testing.assert_array_equal(geom.reshape(2, -1, 3)[:, sel].numpy(), f["geom_sample"])
    np.testing.assert_array_equal(idx.reshape(2, -1, 3)[:, sel].numpy(), f["idx_sample"])
    inr = ((idx[0] >= 0) & (idx[0] < vn.int())).all(-1).reshape(-1).numpy()
    np.testing.assert_array_equal(np.packbits(inr), f["idx_ref_packed"])
    assert int(inr.sum()) == 130108          # 25.9 % of 501,760
    hist = np.zeros((21, 21), np.int64)
    ii = idx[0].reshape(-1, 3).numpy()[inr]
    np.add.at(hist, (ii[:, 1], ii[:, 0]), 1)
    np.testing.assert_array_equal(hist, f["cell_hist"][0])


def test_c_abi_library_exports_every_declared_symbol():
    from thinktwice_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "thinktwice_hip.h")).read()
    names = set(re.findall(r"\b(tt_[a-z0-9_]+)\s*\(", hdr))
    assert "tt_voxel_pool_fwd" in names
    L = _lib.lib()
    for n in sorted(names):
        assert hasattr(L, n), f"libthinktwice_hip.so does not export {n}"
    assert L.tt_version() >= 100


def test_product_op_refuses_cpu_tensors():
    from thinktwice_amd import _lib
    from thinktwice_amd.voxel_pooling import voxel_pooling
    geom = torch.zeros(1, 4, 3, dtype=torch.int32)
    feats = torch.zeros(1, 4, 8)
    with pytest.raises(_lib.TTError):
        voxel_pooling(geom, feats, torch.tensor([21, 21, 1]))


def test_import_path_dropins_expose_the_reference_names():
    """open_loop_training/ops/voxel_pooling/__init__.py:1-3 and the `voxel_pooling_ext` module of voxel_pooling.py:5:
    the same names from a package laid out like the reference's (VERDICT r4 missing #3)."""
    import importlib
    pkg = importlib.import_module("thinktwice_amd.dropin.ops.voxel_pooling")
    assert pkg.__all__ == ["voxel_pooling"] and callable(pkg.voxel_pooling)
    ext = importlib.import_module("thinktwice_amd.dropin.ops.voxel_pooling.voxel_pooling_ext")
    assert callable(ext.voxel_pooling_forward_wrapper)
    import inspect
    assert list(inspect.signature(ext.voxel_pooling_forward_wrapper).parameters) == [
        "batch_size", "num_points", "num_channels", "num_voxel_x", "num_voxel_y", "num_voxel_z", "geom_xyz",
        "input_features", "output_features", "pos_memo"]             # voxel_pooling_forward.cpp:24-37


# ----------------------------------------------------------------------------- GPU parity
def _hip_pool(geom, feats, voxel_num, requires_grad=False):
    from thinktwice_amd.voxel_pooling import voxel_pooling
    g = torch.from_numpy(np.ascontiguousarray(geom)).cuda()
    f = torch.from_numpy(np.ascontiguousarray(feats)).cuda().requires_grad_(requires_grad)
    out = voxel_pooling(g, f, torch.as_tensor(voxel_num))
    return out, f


@pytest.mark.gpu
def test_hip_voxel_pool_matches_golden(golden_dir):
    f = _load(golden_dir, "f3_voxel_pool.npz")
    out, feats = _hip_pool(f["geom"], f["feats"], f["voxel_num"], requires_grad=True)
    np.testing.assert_allclose(out.detach().cpu().numpy(), f["out"], rtol=1e-4, atol=1e-4)
    out.backward(torch.from_numpy(f["grad_out"]).cuda())
    np.testing.assert_array_equal(feats.grad.cpu().numpy(), f["grad_in"])


@pytest.mark.gpu
@pytest.mark.parametrize("B,Np,C", [(1, 1, 4), (2, 63, 256), (3, 1000, 256), (2, 777, 7), (1, 300, 1024), (2, 20000, 256)])
def test_b1_symbol_tt_voxel_pool_fwd_directly(B, Np, C, golden_dir):
    """The B1 boundary symbol ITSELF (`tt_voxel_pool_fwd`, the argument-for-argument mirror of
    voxel_pooling_forward.cpp:24-37 + a stream; the single-pass kernel of voxel_pooling_forward_cuda.cu:9-36) -- not the
    workspace variant the Python operator prefers: sums vs the C oracle (f32 atomics: order differs, values to 1e-4),
    pos_memo bit-exact, out-of-range points leave their -1, and the output is ACCUMULATED into (the reference's
    atomicAdd into a caller-zeroed buffer)."""
    import ctypes
    from thinktwice_amd import _lib
    rng = np.random.default_rng(B * 77 + Np + C)
    geom = np.stack([rng.integers(-2, 23, (B, Np)), rng.integers(-2, 23, (B, Np)),
                     rng.integers(-1, 2, (B, Np))], -1).astype(np.int32)
    feats = rng.standard_normal((B, Np, C), dtype=np.float32)
    ref, memo = c_ref.voxel_pool_fwd(geom, feats, (21, 21, 1))
    g, f = torch.from_numpy(geom).cuda(), torch.from_numpy(feats).cuda()
    out = torch.full((B, 21, 21, C), 0.5, device="cuda")
    pm = torch.full((B, Np, 3), -1, dtype=torch.int32, device="cuda")
    ci = ctypes.c_int
    _lib.check(_lib.lib().tt_voxel_pool_fwd(ci(B), ci(Np), ci(C), ci(21), ci(21), ci(1), _lib.ptr(g), _lib.ptr(f),
                                            _lib.ptr(out), _lib.ptr(pm), _lib.cur_stream(out.device)), "tt_voxel_pool_fwd")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy() - 0.5, ref, rtol=1e-4, atol=2e-4)
    np.testing.assert_array_equal(pm.cpu().numpy(), memo)
    # the import-path drop-in (reference: `from ops.voxel_pooling import voxel_pooling`) runs the same operator
    from thinktwice_amd.dropin.ops.voxel_pooling import voxel_pooling
    from thinktwice_amd.dropin.ops.voxel_pooling import voxel_pooling_ext
    got = voxel_pooling(g, f, torch.tensor([21, 21, 1]))
    np.testing.assert_allclose(got.permute(0, 2, 3, 1).cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    out2 = torch.zeros(B, 21, 21, C, device="cuda")
    pm2 = torch.full((B, Np, 3), -1, dtype=torch.int32, device="cuda")
    vn = torch.tensor([21, 21, 1], device="cuda")                    # 0-dim tensor arguments, like voxel_pooling.py:45-47
    assert voxel_pooling_ext.voxel_pooling_forward_wrapper(B, Np, C, vn[0], vn[1], vn[2], g, f, out2, pm2) == 1
    np.testing.assert_allclose(out2.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(pm2.cpu().numpy(), memo)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Np,C", [(1, 1, 4), (2, 63, 256), (2, 64, 256), (3, 1000, 256),
                                    (1, 4097, 512), (2, 777, 7), (1, 300, 1024), (2, 128, 36),
                                    (2, 20000, 256), (1, 9000, 64), (3, 8193, 1024)])
def test_hip_voxel_pool_matches_oracle_random(B, Np, C):
    rng = np.random.default_rng(B * 1000 + Np + C)
    geom = np.stack([rng.integers(-2, 23, (B, Np)), rng.integers(-2, 23, (B, Np)),
                     rng.integers(-1, 2, (B, Np))], -1).astype(np.int32)
    feats = rng.standard_normal((B, Np, C), dtype=np.float32)
    ref, memo = c_ref.voxel_pool_fwd(geom, feats, (21, 21, 1))
    out, _ = _hip_pool(geom, feats, (21, 21, 1))
    np.testing.assert_allclose(out.permute(0, 2, 3, 1).cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Np,C", [(1, 1, 4), (2, 63, 256), (3, 1000, 256), (1, 4097, 512), (2, 20000, 256),
                                    (1, 9000, 64), (3, 8193, 1024)])
def test_planned_voxel_pool_matches_oracle_random(B, Np, C):
    """Static-geometry plan (tt_voxel_pool_plan_build + tt_voxel_pool_fwd_planned): same sums as the oracle, bit-identical
    between two calls (no atomics), and it accumulates into a pre-filled output like the reference kernel does."""
    from thinktwice_amd.voxel_pooling import VoxelPoolPlan
    rng = np.random.default_rng(B * 1000 + Np + C)
    geom = np.stack([rng.integers(-2, 23, (B, Np)), rng.integers(-2, 23, (B, Np)),
                     rng.integers(-1, 2, (B, Np))], -1).astype(np.int32)
    if Np > 5000:
        geom[0, 100:4000] = (5, 7, 0)                 # one long run: many segments of a single cell
    feats = rng.standard_normal((B, Np, C), dtype=np.float32)
    ref, _ = c_ref.voxel_pool_fwd(geom, feats, (21, 21, 1))
    plan = VoxelPoolPlan(torch.from_numpy(geom).cuda(), (21, 21, 1))
    fd = torch.from_numpy(feats).cuda()
    a = plan(fd)
    b = plan(fd)
    torch.cuda.synchronize()
    np.testing.assert_allclose(a.permute(0, 2, 3, 1).cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    assert torch.equal(a, b)
    pre = torch.full((B, 21, 21, C), 2.0, device="cuda")
    plan.forward_into(fd, pre)
    np.testing.assert_allclose(pre.cpu().numpy(), ref + 2.0, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Np,C,grid,mode", [
    (2, 20000, 256, (21, 21, 1), "random"),          # random cells: up to 64 ballots per slice, ragged last chunk
    (1, 70001, 256, (21, 21, 1), "frustum"),         # runs of equal cells like a frustum, 9 chunks, Np not a multiple of 64
    (3, 16385, 64, (21, 21, 1), "frustum"),          # a chunk of ONE point per sample
    (2, 40000, 512, (32, 32, 1), "random"),          # 1024 cells (the path's limit), C > 256 (NV = 4 kernels)
    (4, 9000, 128, (21, 21, 2), "onecell"),          # every in-range point in one cell; z range 2
    (8, 8192, 256, (21, 21, 1), "allout"),           # nothing in range
])
def test_counting_sort_path_matches_oracle_and_the_static_plan(B, Np, C, grid, mode):
    """Round 5: `tt_voxel_pool_fwd_ws` sorts the in-range points by (sample, cell) PER LAUNCH (counting sort: vp_cs_count /
    scan / scatter) and runs the planned streaming kernels, for any geometry with <= 1024 cells and <= 524,288 points per
    sample.  Sums vs the C oracle, pos_memo bit-exact, bit-identical to the static plan (same row order inside a cell) and to
    itself, accumulation into a pre-filled output; with a workspace that is too small the call still answers (older paths)."""
    import ctypes
    from thinktwice_amd import _lib
    from thinktwice_amd.voxel_pooling import VoxelPoolPlan
    X, Y, Z = grid
    rng = np.random.default_rng(B * 131 + Np + C)
    if mode == "random":
        geom = np.stack([rng.integers(-2, X + 2, (B, Np)), rng.integers(-2, Y + 2, (B, Np)), rng.integers(-1, Z + 1, (B, Np))], -1)
    elif mode == "frustum":
        run = rng.integers(1, 90, (B, Np // 8 + 1))                       # cells change every 1-90 points
        cx = np.repeat(rng.integers(-6, X + 6, run.shape), 8, axis=1)[:, :Np]
        cy = np.repeat(rng.integers(-6, Y + 6, run.shape), 8, axis=1)[:, :Np]
        geom = np.stack([cx, cy, np.zeros_like(cx)], -1)
    elif mode == "onecell":
        geom = np.stack([np.full((B, Np), 3), np.full((B, Np), 17), rng.integers(-1, Z + 1, (B, Np))], -1)
    else:
        geom = np.full((B, Np, 3), -5)
    geom = np.ascontiguousarray(geom.astype(np.int32))
    feats = rng.standard_normal((B, Np, C), dtype=np.float32)
    ref, memo = c_ref.voxel_pool_fwd(geom, feats, grid)
    L, ci = _lib.lib(), ctypes.c_int
    g, f = torch.from_numpy(geom).cuda(), torch.from_numpy(feats).cuda()
    nbytes = int(L.tt_voxel_pool_workspace_bytes(ci(B), ci(Np), ci(C), ci(X), ci(Y)))
    assert nbytes > 0

    def run(ws_bytes, prefill=0.0):
        out = torch.full((B, Y, X, C), prefill, device="cuda")
        pm = torch.full((B, Np, 3), -1, dtype=torch.int32, device="cuda")
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device="cuda")
        _lib.check(L.tt_voxel_pool_fwd_ws(ci(B), ci(Np), ci(C), ci(X), ci(Y), ci(Z), _lib.ptr(g), _lib.ptr(f), _lib.ptr(out),
                                          _lib.ptr(pm), _lib.ptr(ws), ctypes.c_longlong(ws_bytes), _lib.cur_stream(out.device)),
                   "tt_voxel_pool_fwd_ws")
        torch.cuda.synchronize()
        return out, pm

    a, pm = run(nbytes)
    np.testing.assert_allclose(a.cpu().numpy(), ref, rtol=1e-4, atol=2e-4)
    np.testing.assert_array_equal(pm.cpu().numpy(), memo)
    b, _ = run(nbytes)
    assert torch.equal(a, b), "two launches differ: the path must be deterministic"
    planned = VoxelPoolPlan(g, grid)(f).permute(0, 2, 3, 1)
    assert torch.equal(a, planned.contiguous()), "the per-launch sort and the static plan add the same rows in the same order"
    c, _ = run(nbytes, prefill=1.5)
    np.testing.assert_allclose(c.cpu().numpy(), ref + 1.5, rtol=1e-4, atol=2e-4)
    d, pm2 = run(1024)                                   # workspace too small for any two-phase path: single-pass kernel
    np.testing.assert_allclose(d.cpu().numpy(), ref, rtol=1e-4, atol=5e-4)
    np.testing.assert_array_equal(pm2.cpu().numpy(), memo)


@pytest.mark.gpu
def test_hip_voxel_pool_edge_cases():
    from thinktwice_amd.voxel_pooling import voxel_pooling, voxel_pooling_forward_wrapper
    # empty input
    out = voxel_pooling(torch.zeros(2, 0, 3, dtype=torch.int32).cuda(), torch.zeros(2, 0, 256).cuda(),
                        torch.tensor([21, 21, 1]))
    assert out.shape == (2, 256, 21, 21) and float(out.abs().sum()) == 0.0
    # everything out of range
    geom = torch.full((1, 500, 3), -1, dtype=torch.int32).cuda()
    out = voxel_pooling(geom, torch.ones(1, 500, 256).cuda(), torch.tensor([21, 21, 1]))
    assert float(out.abs().sum()) == 0.0
    # every point in ONE cell (maximum contention), exact integer sums
    geom = torch.zeros(1, 5000, 3, dtype=torch.int32).cuda()
    geom[..., 0] = 20
    geom[..., 1] = 20
    out = voxel_pooling(geom, torch.ones(1, 5000, 256).cuda(), torch.tensor([21, 21, 1]))
    assert float(out[0, :, 20, 20].min()) == 5000.0 and float(out.sum()) == 5000.0 * 256
    # extension-symbol mirror writes pos_memo like the reference kernel (cu:28-30)
    memo = torch.full((1, 5000, 3), -1, dtype=torch.int32).cuda()
    o = torch.zeros(1, 21, 21, 256).cuda()
    assert voxel_pooling_forward_wrapper(1, 5000, 256, 21, 21, 1, geom, torch.ones(1, 5000, 256).cuda(),
                                         o, memo) == 1
    assert (memo.cpu() == torch.tensor([0, 20, 20], dtype=torch.int32)).all()
    # non-contiguous input is rejected like the reference (voxel_pooling.py:25-26)
    with pytest.raises(AssertionError):
        voxel_pooling(geom, torch.ones(1, 256, 5000).cuda().permute(0, 2, 1), torch.tensor([21, 21, 1]))


@pytest.mark.gpu
def test_hip_frustum_index_bit_exact_and_full_size_pool():
    """BASELINE.json full size: 4 cams x 80 x 28 x 56 points, C=256, real calibration."""
    from thinktwice_amd import ops, synth
    fr = og.create_frustum()
    vs, vc, vn = og.voxel_constants()
    metas = synth.make_img_metas(2)
    intr, ida, s2e, _, _ = og.assemble_camera_mats(metas)
    inv_ida, comb = og.geometry_mats(s2e[:, -1], intr[:, -1], ida[:, -1])
    mats = torch.stack([inv_ida, comb], 2).reshape(-1, 2, 4, 4).contiguous().cuda()
    lo = (vc - vs / 2.0).tolist()
    geom, gf = ops.frustum_voxel_index(fr.cuda(), mats, lo, vs.tolist(), 2, 4, want_f32=True)
    geom_o = og.get_geometry(fr, s2e[:, -1], intr[:, -1], ida[:, -1])
    idx_o = og.voxel_index(geom_o, vc, vs).reshape(2, -1, 3)
    np.testing.assert_array_equal(gf.cpu().numpy(), geom_o.reshape(2, -1, 3).numpy())
    np.testing.assert_array_equal(geom.cpu().numpy(), idx_o.numpy())       # bit exact (integer work)
    # full-size pool vs oracle (B=2 x 501,760 points x 256 channels)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(2, geom.shape[1], 256, generator=g)
    ref, _ = c_ref.voxel_pool_fwd(idx_o.numpy(), feats.numpy(), vn.tolist())
    from thinktwice_amd.voxel_pooling import voxel_pooling
    out = voxel_pooling(geom, feats.cuda(), vn)
    got = out.permute(0, 2, 3, 1).cpu().numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 1e-5, err
    # size-independent property: total mass is conserved (sum over cells == sum over kept points)
    inr = ((idx_o >= 0) & (idx_o < vn.int())).all(-1)
    want = feats.double()[inr].sum(0)
    np.testing.assert_allclose(out.double().sum((0, 2, 3)).cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-2)
    # the static-geometry plan of the same (real) geometry gives the same BEV
    from thinktwice_amd.voxel_pooling import VoxelPoolPlan
    planned = VoxelPoolPlan(geom, vn)(feats.cuda()).permute(0, 2, 3, 1).cpu().numpy()
    assert np.abs(planned - ref).max() / np.abs(ref).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_hip_fused_lift_splat_matches_materialised_reference_formulation(dt):
    """Fused kernel == softmax (x) context outer product -> permute -> voxel pooling (lss.py:583-632)."""
    from thinktwice_amd import ops
    B, N, D, H, W, C = 2, 4, 80, 7, 9, 64
    g = torch.Generator().manual_seed(77)
    depth = torch.randn(B * N, D, H, W, generator=g) * 2
    ctx = torch.randn(B * N, C, H, W, generator=g)
    geom = torch.stack([torch.randint(-2, 23, (B, N * D * H * W), generator=g),
                        torch.randint(-2, 23, (B, N * D * H * W), generator=g),
                        torch.randint(-1, 2, (B, N * D * H * W), generator=g)], -1).to(torch.int32)
    # channel-last device inputs
    d_cl = depth.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    c_cl = ctx.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    out = ops.lift_splat(d_cl, c_cl, geom.cuda(), (21, 21, 1), B, N)            # [B,Y,X,C]
    out_rf = ops.lift_splat(d_cl, c_cl, geom.cuda(), (21, 21, 1), B, N, rot_flip=True)
    # oracle: the reference's materialised formulation on the same (rounded) inputs
    dq = d_cl.float().cpu().permute(0, 3, 1, 2)
    cq = c_cl.float().cpu().permute(0, 3, 1, 2)
    vol = dq.softmax(1).unsqueeze(1) * cq.unsqueeze(2)                            # [BN,C,D,H,W]
    vol = vol.reshape(B, N, C, D, H, W).permute(0, 1, 3, 4, 5, 2).contiguous()
    ref, _ = c_ref.voxel_pool_fwd(geom.numpy(), vol.reshape(B, -1, C).numpy(), (21, 21, 1))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-4, atol=2e-5)
    bev = torch.from_numpy(ref).permute(0, 3, 1, 2)                              # [B,C,Y,X]
    want = torch.rot90(torch.flip(bev, dims=[2]), 1, dims=[2, 3])                # EDF:241
    np.testing.assert_allclose(out_rf.permute(0, 3, 1, 2).cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.gpu
def test_lift_splat_workspace_form_is_bit_reproducible_and_matches_the_atomic_form(monkeypatch):
    """tt_lift_splat_fwd_ws (partial rows per (strip, cell) + ordered per-cell reduce, no f32 atomics) on ray-like geometry
    (every image column falls into a handful of BEV cells, as with a real frustum): equal to the materialised formulation,
    to the atomic single-kernel form within rounding, and bit-identical from run to run."""
    from thinktwice_amd import ops
    B, N, D, H, W, C, X = 2, 4, 80, 16, 24, 80, 32
    g = torch.Generator().manual_seed(5)
    depth = (torch.randn(B * N, H, W, D, generator=g) * 2).cuda()
    ctx = torch.randn(B * N, H, W, C, generator=g).cuda()
    # rays: azimuth from (camera, column), range from the depth bin; a few rows look over the grid's edge
    cam = torch.arange(N).view(1, N, 1, 1, 1).float()
    d = torch.arange(D).view(1, 1, D, 1, 1).float()
    h = torch.arange(H).view(1, 1, 1, H, 1).float()
    w = torch.arange(W).view(1, 1, 1, 1, W).float()
    b = torch.arange(B).view(B, 1, 1, 1, 1).float()
    az = cam * (np.pi / 2) + (w / W - 0.5) * 1.4 + 0.05 * b
    rng = 0.5 + d * 0.25 + 0.0 * h
    gx = torch.floor(X / 2 + rng * torch.cos(az)).expand(B, N, D, H, W)
    gy = torch.floor(X / 2 + rng * torch.sin(az)).expand(B, N, D, H, W)
    gz = torch.where(h.expand(B, N, D, H, W) == H - 1, 1.0, 0.0)
    geom = torch.stack([gx, gy, gz], -1).reshape(B, N * D * H * W, 3).to(torch.int32).cuda()
    out1 = ops.lift_splat(depth, ctx, geom, (X, X, 1), B, N)
    out2 = ops.lift_splat(depth, ctx, geom, (X, X, 1), B, N)
    assert torch.equal(out1, out2)
    monkeypatch.setattr(ops, "_LIFT_SPLAT_ATOMIC", True)
    out_at = ops.lift_splat(depth, ctx, geom, (X, X, 1), B, N)
    np.testing.assert_allclose(out1.cpu().numpy(), out_at.cpu().numpy(), rtol=2e-5, atol=2e-5)
    vol = depth.cpu().permute(0, 3, 1, 2).softmax(1).unsqueeze(1) * ctx.cpu().permute(0, 3, 1, 2).unsqueeze(2)
    vol = vol.reshape(B, N, C, D, H, W).permute(0, 1, 3, 4, 5, 2).contiguous()
    ref, _ = c_ref.voxel_pool_fwd(geom.cpu().numpy(), vol.reshape(B, -1, C).numpy(), (X, X, 1))
    np.testing.assert_allclose(out1.cpu().numpy(), ref, rtol=2e-4, atol=2e-5)
    assert float(out1.abs().sum()) > 0
