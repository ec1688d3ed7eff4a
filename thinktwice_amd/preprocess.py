"""GPU camera preprocessing -- host-side mirror of the eval image pipeline
(IDAImageTransform is_train=False + img_transform + ImageTransformMulti aug=False:
open_loop_training/code/datasets/pipelines/transform.py:222-378,140-166), one fused kernel."""
import ctypes

import torch

from . import _lib, calib, ops
from .ops import _c, check, lib, ptr


class ImagePreprocessor:
    def __init__(self, final_dim=(calib.FINAL_H, calib.FINAL_W), device="cuda", undistort=True):
        self.device = torch.device(device)
        self.final_dim = tuple(final_dim)
        H, W = calib.IMG_H, calib.IMG_W
        fh, fw = self.final_dim
        resize = max(fh / H, fw / W)                       # sample_ida_augmentation, eval branch (transform.py:264-273)
        self.resized = (int(H * resize), int(W * resize))
        self.crop = (self.resized[0] - fh, int(max(0, self.resized[1] - fw) / 2))
        if undistort:
            mx, my = calib.undistort_rectify_map(W, H)
        else:   # identity map: pixel centres (the -0.5 of align_corners=False is applied in the kernel)
            import numpy as np
            mx = np.broadcast_to(np.arange(W, dtype=np.float32)[None] + 0.5, (H, W)).copy()
            my = np.broadcast_to(np.arange(H, dtype=np.float32)[:, None] + 0.5, (H, W)).copy()
        self.mapx = torch.from_numpy(mx).to(self.device).contiguous()
        self.mapy = torch.from_numpy(my).to(self.device).contiguous()
        self.mean = (ctypes.c_float * 3)(*calib.IMAGENET_MEAN)
        self.std = (ctypes.c_float * 3)(*calib.IMAGENET_STD)

    def __call__(self, raw, channel_last_dtype=None, c_pad=None):
        """raw uint8 (..., 900, 1600, 3) on device -> f32 (..., 3, fh, fw) like the reference pipeline, or, with
        `channel_last_dtype`, the channel-last padded tensor (NI, fh, fw, c_pad) the LSS trunk consumes."""
        _lib.require_cuda(raw)
        assert raw.dtype == torch.uint8 and raw.shape[-1] == 3 and raw.is_contiguous()
        lead = raw.shape[:-3]
        H, W = raw.shape[-3], raw.shape[-2]
        NI = 1
        for d in lead:
            NI *= d
        fh, fw = self.final_dim
        nchw = nhwc = None
        if channel_last_dtype is None:
            nchw = torch.empty(NI, 3, fh, fw, dtype=torch.float32, device=raw.device)
            cp, code = 3, _lib.TT_F32
        else:
            cp = c_pad or (4 if channel_last_dtype == torch.float32 else 8)
            nhwc = torch.empty(NI, fh, fw, cp, dtype=channel_last_dtype, device=raw.device)
            code = ops.dtype_code(nhwc)
        check(lib().tt_preprocess_images(ptr(raw), _c(NI), _c(H), _c(W), ptr(self.mapx), ptr(self.mapy),
                                         _c(self.resized[0]), _c(self.resized[1]), _c(self.crop[0]), _c(self.crop[1]),
                                         _c(fh), _c(fw), self.mean, self.std, ptr(nhwc), _c(cp), _c(code), ptr(nchw),
                                         ops.cur_stream(raw.device)), "tt_preprocess_images")
        if nchw is not None:
            return nchw.view(*lead, 3, fh, fw)
        return nhwc
