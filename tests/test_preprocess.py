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
