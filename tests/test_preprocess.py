"""Fused GPU camera preprocessing vs the torch-CPU restatement of the reference pipeline."""
import numpy as np
import pytest
import torch

from thinktwice_amd import calib


def test_undistort_map_is_sane():
    mx, my = calib.undistort_rectify_map()
    assert mx.shape == (900, 1600) and mx.dtype == np.float32
    # principal point of the new camera maps (almost) onto the raw principal point; map is monotone
    assert abs(mx[450, 788] - 800) < 1.0 and abs(my[450, 788] - 450) < 1.0
    assert (np.diff(mx[450]) > 0).all() and (np.diff(my[:, 800]) > 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("undistort", [True, False])
def test_preprocess_matches_oracle(undistort):
    from oracle import preprocess_ref
    from thinktwice_amd.preprocess import ImagePreprocessor
    g = torch.Generator().manual_seed(3)
    raw = torch.randint(0, 256, (3, 900, 1600, 3), generator=g, dtype=torch.uint8)
    # smooth structure on top of noise so that interpolation errors would show
    yy = torch.arange(900).view(1, 900, 1, 1)
    raw = ((raw.float() * 0.25) + (yy % 200).float() * 0.9).clamp(0, 255).to(torch.uint8)
    pp = ImagePreprocessor(undistort=undistort)
    ref = preprocess_ref.preprocess(raw, pp.mapx.cpu(), pp.mapy.cpu())
    out = pp(raw.cuda())
    assert out.shape == (3, 3, 448, 896)
    err = float((out.cpu() - ref).abs().max())
    mean_err = float((out.cpu() - ref).abs().mean())
    print("preprocess max abs err", err, "mean", mean_err)
    # the reference normalises the map to [-1,1] in f32 and grid_sample un-normalises it again: ~1e-4 px of
    # coordinate rounding on a 1600-px axis times the (noisy) image gradient bounds the achievable agreement
    assert err < 2e-3 and mean_err < 2e-5, (err, mean_err)
    cl = pp(raw.cuda(), channel_last_dtype=torch.float32)
    assert cl.shape == (3, 448, 896, 4) and float(cl[..., 3].abs().max()) == 0.0
    assert torch.equal(cl[..., :3].permute(0, 3, 1, 2), out)


def _f17():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f17_image_pipeline.npz"))


def _check_against_f17(out, tol_max, tol_mean):
    """out: f32 array [T, N, 3, 448, 896] -> errors against the reference pipeline's own outputs (golden F17)."""
    g = _f17()
    samp = out.reshape(-1)[g["sample_idx"]]
    errs = [np.abs(samp - g["sample_val"]), np.abs(out[-1, :, :, 200, :] - g["row_200"]),
            np.abs(out[-1, :, :, :, 431] - g["col_431"])]
    emax = max(float(e.max()) for e in errs)
    emean = max(float(e.mean()) for e in errs)
    assert emax <= tol_max and emean <= tol_mean, (emax, emean)
    np.testing.assert_allclose(out.mean(axis=(2, 3, 4)), g["per_image_mean"], atol=max(tol_mean, 1e-6) * 2)
    return emax, emean


def test_oracle_preprocess_matches_reference_pipeline_golden_f17():
    """Pins oracle/preprocess_ref.py (VERDICT r3 missing #1): golden F17 = the reference's own IDAImageTransform.__call__ +
    img_transform + ImageTransformMulti(aug=False) (transform.py:275-341, 346-378, 144-163) on two seeded uint8 sweeps."""
    from oracle import preprocess_ref
    from thinktwice_amd import synth
    raw = synth.raw_camera_frames(seed=17)
    assert int(_f17()["seed"][0]) == 17
    mx, my = calib.undistort_rectify_map()
    with torch.no_grad():
        out = preprocess_ref.preprocess(torch.from_numpy(raw.reshape(-1, 900, 1600, 3)), mx, my).numpy()
    emax, emean = _check_against_f17(out.reshape(2, 4, 3, 448, 896), 1e-5, 1e-7)   # same torch ops: agreement to rounding
    print("oracle vs F17: max", emax, "mean", emean)


def test_calibration_tables_match_reference_pipeline_golden_f17():
    """ida_mats / cam_intrinsic / lidar2img / lidar2cam as IDAImageTransform attaches them to img_metas (transform.py:333-340)."""
    g = _f17()
    intr, l2c, l2i = calib.camera_tables()
    np.testing.assert_array_equal(g["cam_intrinsic"], intr)
    np.testing.assert_array_equal(g["lidar2cam"], l2c)
    np.testing.assert_array_equal(g["lidar2img"], l2i)
    ida = calib.eval_ida_mat()
    assert g["ida_mats"].shape == (2, 4, 4, 4)
    for t in range(2):
        for n in range(4):
            np.testing.assert_array_equal(g["ida_mats"][t, n], ida)


@pytest.mark.gpu
def test_gpu_preprocess_matches_reference_pipeline_golden_f17():
    """tt_preprocess_images against the reference pipeline's outputs directly (not only against the oracle)."""
    from thinktwice_amd import synth
    from thinktwice_amd.preprocess import ImagePreprocessor
    raw = torch.from_numpy(synth.raw_camera_frames(seed=17))
    out = ImagePreprocessor()(raw.cuda()).cpu().numpy()
    assert out.shape == (2, 4, 3, 448, 896)
    # same bound as against the oracle (the reference round-trips the map through its [-1, 1] f32 normalisation)
    emax, emean = _check_against_f17(out, 2e-3, 2e-5)
    print("tt_preprocess_images vs F17: max", emax, "mean", emean)
