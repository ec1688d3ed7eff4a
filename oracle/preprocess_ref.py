"""Oracle (TEST INFRASTRUCTURE): eval image pipeline of the reference restated with torch CPU ops
(datasets/pipelines/transform.py: IDAImageTransform.__call__ :283-286, img_transform :346-356 (T.Resize on
a float tensor == F.interpolate bilinear, align_corners=False, no antialias in torchvision 0.13),
ImageTransformMulti aug=False :163).  PINNED by golden F17 (tests/golden/gen_golden.py::gen_f17 runs the reference's own
transform.py on two seeded sweeps; tests/test_preprocess.py holds this file to it at 1e-5).  The undistortion TABLE comes
from cv2 in the reference; OpenCV is absent here, so `thinktwice_amd.calib.undistort_rectify_map` restates the published
pinhole + Brown-Conrady model (that table alone stays unpinned; everything downstream of it is pinned)."""
import torch
import torch.nn.functional as F


def preprocess(raw_u8, mapx, mapy, final_dim=(448, 896)):
    """raw_u8 (NI, H, W, 3) uint8 -> (NI, 3, fh, fw) f32."""
    NI, H, W, _ = raw_u8.shape
    img = raw_u8.to(torch.float32).permute(0, 3, 1, 2)
    gx = (torch.as_tensor(mapx) - W / 2) / (W / 2)
    gy = (torch.as_tensor(mapy) - H / 2) / (H / 2)
    grid = torch.stack([gx, gy], -1).unsqueeze(0).repeat(NI, 1, 1, 1)
    und = F.grid_sample(img, grid, align_corners=False)
    fh, fw = final_dim
    resize = max(fh / H, fw / W)
    rw, rh = int(W * resize), int(H * resize)
    crop_h = rh - fh
    crop_w = int(max(0, rw - fw) / 2)
    res = F.interpolate(und, size=(rh, rw), mode="bilinear", align_corners=False)
    res = res[..., crop_h:crop_h + fh, crop_w:crop_w + fw]
    x = res / 255.0
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return (x - mean) / std
