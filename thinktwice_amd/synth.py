"""Seeded synthetic inputs with the shapes/dtypes of the reference's `forward_inference`
batch (input contract: open_loop_training/code/datasets/carla_dataset.py:263-334,
leaderboard/team_code/thinktwice_agent.py:444-454).  SURVEY.md section 8(d)."""
import numpy as np
import torch

from . import calib


def make_img_metas(batch_size, num_sweeps=2, curr2key=None, final_dim=(calib.FINAL_H, calib.FINAL_W), jitter_seed=None):
    """img_metas[b][t] dicts with the keys LSS.forward reads (backbones/lss.py:667-707).

    `jitter_seed`: per (sample, camera) perturbed calibration -- focal lengths x (1 +- 8 %), principal point +- 20 px, the
    image-augmentation matrix's resize x (1 +- 5 %) and crop offsets +- 6 px (what the reference's training-time
    IDAImageTransform.sample_ida_augmentation does to `ida_mats`, transform.py:250-263), lidar2img rebuilt from the perturbed
    intrinsics.  The train-mode goldens (F11 / F16) use it: with the fixed evaluation rig DepthNet's BatchNorm1d over the 22
    camera parameters (lss.py:206-231) has zero-variance columns of magnitude ~800, whose batch-statistics output is
    beta + rounding noise of order ulp(800 / sqrt(eps)) ~ 0.01 that depends on the implementation's operation order (torch's
    CPU kernel folds the mean into the shift) -- not a property any second implementation can reproduce."""
    intr0, l2c, l2i0 = calib.camera_tables()
    ida0 = np.stack([calib.eval_ida_mat(final_dim) for _ in range(4)])
    rng = None if jitter_seed is None else np.random.default_rng(jitter_seed)
    metas = []
    for _ in range(batch_size):
        intr, ida, l2i = intr0, ida0, l2i0
        if rng is not None:
            intr, ida = intr0.copy(), ida0.copy()
            for c in range(4):
                intr[c, 0, 0] *= 1.0 + 0.08 * rng.uniform(-1, 1)
                intr[c, 1, 1] *= 1.0 + 0.08 * rng.uniform(-1, 1)
                intr[c, 0, 2] += 20.0 * rng.uniform(-1, 1)
                intr[c, 1, 2] += 20.0 * rng.uniform(-1, 1)
                s = 1.0 + 0.05 * rng.uniform(-1, 1)
                ida[c, 0, 0] *= s
                ida[c, 1, 1] *= s
                ida[c, 0, 3] += 6.0 * rng.uniform(-1, 1)
                ida[c, 1, 3] += 6.0 * rng.uniform(-1, 1)
            k4 = np.tile(np.eye(4, dtype=np.float32), (4, 1, 1))
            k4[:, :3, :3] = intr
            l2i = (k4 @ l2c).astype(np.float32)
        per_sweep = []
        for t in range(num_sweeps):
            c2k = np.eye(4, dtype=np.float32)
            if curr2key is not None and t < num_sweeps - 1:
                c2k = np.asarray(curr2key, dtype=np.float32)
            per_sweep.append({
                "cam_intrinsic": torch.from_numpy(intr.copy()),
                "ida_mats": torch.from_numpy(ida.copy()),
                # carla_dataset.py:290,312: lidar2cam @ curr2key (key frame: identity)
                "currlidar2keycam": torch.from_numpy(l2c @ c2k),
                "lidar2cam": torch.from_numpy(l2c.copy()),
                "lidar2img": torch.from_numpy(l2i.copy()),
            })
        metas.append(per_sweep)
    return metas


def make_batch(batch_size, seed=1234, num_points=65536, img_hw=(calib.FINAL_H, calib.FINAL_W),
               device="cpu", with_img=True, jitter_calib=None):
    """Synthetic `forward_inference` batch: img N(0,1) (B,2,4,3,H,W); points (B,1,Np,5)
    uniform in the point-cloud range with z<4; speed/target_point/command."""
    imgs, pts, speed, tp, cmd = [], [], [], [], []
    for i in range(batch_size):
        g = torch.Generator().manual_seed(seed + i)
        if with_img:
            imgs.append(torch.randn(2, 4, 3, img_hw[0], img_hw[1], generator=g))
        u = torch.rand(num_points, 5, generator=g)
        p = torch.empty(num_points, 5)
        p[:, 0] = -8.0 + u[:, 0] * 38.4
        p[:, 1] = -19.2 + u[:, 1] * 38.4
        p[:, 2] = -4.0 + u[:, 2] * 8.0
        p[:, 3] = u[:, 3]
        p[:, 4] = 0.0
        p[num_points // 2:, 4] = -1.0
        pts.append(p[None])
        speed.append(torch.rand(1, generator=g) * 12.0)
        tp.append(torch.randn(2, generator=g) * 10.0)
        c = int(torch.randint(0, 6, (1,), generator=g))
        oh = torch.zeros(6)
        oh[c] = 1.0
        cmd.append(oh)
    batch = {
        "points": torch.stack(pts).to(device),
        "speed": torch.cat(speed).to(device),
        "target_point": torch.stack(tp).to(device),
        "target_command": torch.stack(cmd).to(device),
        "target_command_raw": torch.stack(cmd).argmax(-1).to(device),
        "img_metas": make_img_metas(batch_size, final_dim=img_hw, jitter_seed=jitter_calib),
    }
    if with_img:
        batch["img"] = torch.stack(imgs).to(device)
    return batch


def make_train_targets(batch_size, img_hw=(calib.FINAL_H, calib.FINAL_W), seed=4321, num_cams=4):
    """Synthetic supervision for `forward_train` with the shapes the reference dataset collates
    (carla_dataset.py / thinktwice_decoder.py:536-619, encoder_decoder_framework.py:148-191):
    expert waypoints and Beta action parameters (current + 4 future steps), expert value / flattened feature /
    BEV grid features (Roach maps: 32x21x21, 64x10x10, 128x4x4, 256x2x2 at list indices 2..5), future grid
    features, sparse per-camera depth maps in metres (0 = no return) and 12-class segmentation labels."""
    g = torch.Generator().manual_seed(seed)
    B, (H, W) = batch_size, img_hw
    r = lambda *s: torch.randn(*s, generator=g)                       # noqa: E731
    pos = lambda *s: torch.rand(*s, generator=g) * 4.0 + 0.2          # noqa: E731  Beta parameters > 0
    grid = lambda: [r(B, 1), r(B, 1), r(B, 32, 21, 21), r(B, 64, 10, 10), r(B, 128, 4, 4), r(B, 256, 2, 2)]   # noqa: E731
    depth = torch.rand(B, num_cams, H, W, generator=g) * 45.0
    depth[torch.rand(B, num_cams, H, W, generator=g) < 0.9] = 0.0     # LiDAR-projected depth is sparse
    return {
        "waypoints": r(B, 4, 2) * 3.0,
        "action_mu": pos(B, 2), "action_sigma": pos(B, 2),
        "future_action_mu": [pos(B, 2) for _ in range(4)], "future_action_sigma": [pos(B, 2) for _ in range(4)],
        "value": r(B), "feature": r(B, 256),
        "grid_feature": grid(), "future_grid_feature": [grid() for _ in range(4)],
        "depth": depth,
        "seg": torch.randint(0, 12, (B, num_cams, H, W), generator=g).float(),
    }


def raw_camera_frames(seed=17, T=2, N=4, h=900, w=1600):
    """Seeded raw camera frames uint8 [T, N, h, w, 3] (what the agent hands the image pipeline): noise on top of smooth
    structure, so that interpolation errors would show.  Golden F17 was produced from exactly these frames."""
    import numpy as np
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, (T, N, h, w, 3), dtype=np.uint8).astype(np.float32)
    yy = (np.arange(h, dtype=np.float32)[:, None, None] % 200) * 0.9
    xx = (np.arange(w, dtype=np.float32)[None, :, None] % 320) * 0.35
    return np.clip(noise * 0.25 + yy + xx, 0, 255).astype(np.uint8)
