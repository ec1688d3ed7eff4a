"""Thin tensor->pointer wrappers over the C ABI (include/thinktwice_hip.h).

All functions take device tensors and launch on torch's current stream.  No function here
computes anything in PyTorch: allocation and stream handling only.
"""
import ctypes

import torch

from . import _lib
from ._lib import TT_BF16, TT_F32, check, cur_stream, lib, ptr, require_cuda

_c = ctypes.c_int


def dtype_code(t):
    if t.dtype == torch.float32:
        return TT_F32
    if t.dtype == torch.bfloat16:
        return TT_BF16
    raise _lib.TTError(f"unsupported dtype {t.dtype}")


def frustum_voxel_index(frustum, mats, voxel_lo, voxel_size, batch_size, num_cams, want_f32=False):
    """frustum [D,fH,fW,4] f32 (device); mats [B*ncam,2,4,4] f32 (device): inv(ida), s2e@inv(K).
    voxel_lo / voxel_size: 3 python floats each.  -> geom_xyz int32 [B, ncam*D*fH*fW, 3]."""
    require_cuda(frustum, mats)
    assert frustum.is_contiguous() and mats.is_contiguous()
    D, fH, fW, _ = frustum.shape
    geom = torch.empty(batch_size, num_cams * D * fH * fW, 3, dtype=torch.int32, device=frustum.device)
    gf = torch.empty(geom.shape, dtype=torch.float32, device=frustum.device) if want_f32 else None
    lo = (ctypes.c_float * 3)(*[float(v) for v in voxel_lo])
    sz = (ctypes.c_float * 3)(*[float(v) for v in voxel_size])
    rc = lib().tt_frustum_voxel_index(_c(batch_size), _c(num_cams), _c(D), _c(fH), _c(fW),
                                      ptr(frustum), ptr(mats), lo, sz, ptr(geom), ptr(gf),
                                      cur_stream(frustum.device))
    check(rc, "tt_frustum_voxel_index")
    return (geom, gf) if want_f32 else geom


def lift_splat(depth_logits, context, geom_xyz, voxel_num, batch_size, num_cams, out=None,
               out_coff=0, rot_flip=False):
    """depth_logits [B*ncam,fH,fW,D], context [B*ncam,fH,fW,C] (channel-last, f32 or bf16),
    geom_xyz int32 [B, ncam*D*fH*fW, 3] -> out f32 [B, Y, X, Ctot] (accumulated into `out`)."""
    require_cuda(depth_logits, context, geom_xyz)
    assert depth_logits.is_contiguous() and context.is_contiguous() and geom_xyz.is_contiguous()
    BN, fH, fW, D = depth_logits.shape
    C = context.shape[-1]
    vx, vy, vz = (int(v) for v in voxel_num)
    if out is None:
        oh, ow = (vx, vy) if rot_flip else (vy, vx)
        out = torch.zeros(batch_size, oh, ow, C, dtype=torch.float32, device=context.device)
    rc = lib().tt_lift_splat_fwd(_c(batch_size), _c(num_cams), _c(D), _c(fH), _c(fW), _c(C), _c(vx),
                                 _c(vy), _c(vz), ptr(depth_logits), ptr(context),
                                 _c(dtype_code(context)), ptr(geom_xyz), ptr(out),
                                 _c(out.shape[-1]), _c(out_coff), _c(1 if rot_flip else 0),
                                 cur_stream(context.device))
    check(rc, "tt_lift_splat_fwd")
    return out
