"""CARLA sensor-rig constants used by the reference data pipeline (data, not code).

Values are the calibration tables of
open_loop_training/code/datasets/pipelines/transform.py:24-51 for the camera order
configs/thinktwice.py:102  ['rgb_front', 'rgb_left', 'rgb_right', 'rgb_back'], with
`undistort=True` (configs/thinktwice.py:61 -> newcameramtx / UNDISTORT_LIDAR2IMG).
"""
import numpy as np

CAMERA_NAMES = ["rgb_front", "rgb_left", "rgb_right", "rgb_back"]

# undistorted intrinsics (transform.py:49-51), identical for the 4 cameras
CAM_INTRINSIC = np.array([[304.14395142, 0.0, 788.25758876],
                          [0.0, 221.49429321, 449.78972161],
                          [0.0, 0.0, 1.0]], dtype=np.float64)

LIDAR2CAM = {  # transform.py:25-30
    "rgb_front": [[0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 2.5], [1.0, 0.0, 0.0, -1.5], [0.0, 0.0, 0.0, 1.0]],
    "rgb_back": [[0.0, -1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 2.5], [-1.0, 0.0, 0.0, -1.6], [0.0, 0.0, 0.0, 1.0]],
    "rgb_left": [[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, -1.0, 2.5], [0.0, -1.0, 0.0, -0.3], [0.0, 0.0, 0.0, 1.0]],
    "rgb_right": [[-1.0, 0.0, 0.0, 0.0], [0.0, 0.0, -1.0, 2.5], [0.0, 1.0, 0.0, -0.3], [0.0, 0.0, 0.0, 1.0]],
}

UNDISTORT_LIDAR2IMG = {  # transform.py:33-38
    "rgb_front": [[788.25758876, 304.14395142, 0.0, -1182.38638314], [449.78972161, 0.0, -221.49429321, -120.94884939000008], [1.0, 0.0, 0.0, -1.5], [0.0, 0.0, 0.0, 1.0]],
    "rgb_left": [[304.14395142, -788.25758876, 0.0, -236.47727662799997], [0.0, -449.78972161, -221.49429321, 418.79881654199994], [0.0, -1.0, 0.0, -0.3], [0.0, 0.0, 0.0, 1.0]],
    "rgb_right": [[-304.14395142, 788.25758876, 0.0, -236.47727662799997], [0.0, 449.78972161, -221.49429321, 418.79881654199994], [0.0, 1.0, 0.0, -0.3], [0.0, 0.0, 0.0, 1.0]],
    "rgb_back": [[-788.25758876, -304.14395142, 0.0, -1261.2121420160001], [-449.78972161, 0.0, -221.49429321, -165.9278215510001], [-1.0, 0.0, 0.0, -1.6], [0.0, 0.0, 0.0, 1.0]],
}

IMG_H, IMG_W = 900, 1600          # raw camera size (configs/thinktwice.py:116-117)
FINAL_H, FINAL_W = 448, 896       # network input (configs/thinktwice.py:113)


def eval_ida_mat(final_dim=(FINAL_H, FINAL_W)):
    """IDAImageTransform.sample_ida_augmentation, is_train=False branch (transform.py:264-273)
    followed by img_transform's matrix (transform.py:346-378): resize 0.56, crop rows 56:504."""
    fh, fw = final_dim
    resize = max(fh / IMG_H, fw / IMG_W)
    new_w, new_h = int(IMG_W * resize), int(IMG_H * resize)
    crop_h = int(new_h) - fh
    crop_w = int(max(0, new_w - fw) / 2)
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = m[1, 1] = np.float32(resize)
    m[0, 3] = -float(crop_w)
    m[1, 3] = -float(crop_h)
    return m


def camera_tables():
    """(cam_intrinsic [4,3,3], lidar2cam [4,4,4], lidar2img [4,4,4]) float32, as the reference's
    IDAImageTransform.__init__ builds them (transform.py:241-243)."""
    intr = np.stack([CAM_INTRINSIC for _ in CAMERA_NAMES]).astype(np.float32)
    l2c = np.stack([np.array(LIDAR2CAM[n]) for n in CAMERA_NAMES]).astype(np.float32)
    l2i = np.stack([np.array(UNDISTORT_LIDAR2IMG[n]) for n in CAMERA_NAMES]).astype(np.float32)
    return intr, l2c, l2i
