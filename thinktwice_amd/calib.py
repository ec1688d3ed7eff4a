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


# raw (distorted) pinhole + distortion coefficients [k1, k2, p1, p2, k3] (transform.py:47-48)
RAW_INTRINSIC = np.array([[214.35935394, 0.0, 800.0], [0.0, 214.35935394, 450.0], [0.0, 0.0, 1.0]], dtype=np.float64)
DIST_COEFFS = np.array([0.00888296, -0.00130899, 0.00012061, -0.00338673, 0.00028834], dtype=np.float64)
IMAGENET_MEAN = (0.485, 0.456, 0.406)     # transform.py:144
IMAGENET_STD = (0.229, 0.224, 0.225)


def undistort_rectify_map(width=IMG_W, height=IMG_H):
    """(mapx, mapy) float32 [H, W]: restatement of cv2.initUndistortRectifyMap(mtx, dist, None, newcameramtx,
    (1600, 900), CV_32FC1) as called at transform.py:235 (OpenCV is not installed here: the published pinhole +
    Brown-Conrady model; "parity unpinned" for this table)."""
    fx, fy, cx, cy = RAW_INTRINSIC[0, 0], RAW_INTRINSIC[1, 1], RAW_INTRINSIC[0, 2], RAW_INTRINSIC[1, 2]
    nfx, nfy, ncx, ncy = CAM_INTRINSIC[0, 0], CAM_INTRINSIC[1, 1], CAM_INTRINSIC[0, 2], CAM_INTRINSIC[1, 2]
    k1, k2, p1, p2, k3 = DIST_COEFFS
    u = np.arange(width, dtype=np.float64)[None, :]
    v = np.arange(height, dtype=np.float64)[:, None]
    x = (u - ncx) / nfx + 0 * v
    y = (v - ncy) / nfy + 0 * u
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    two_xy = 2 * x * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + p1 * two_xy + p2 * (r2 + 2 * x2)
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * two_xy
    return (fx * xd + cx).astype(np.float32), (fy * yd + cy).astype(np.float32)
