"""`voxel_pooling(geom_xyz, input_features, voxel_num)` -- host-side mirror of the
reference operator (open_loop_training/ops/voxel_pooling/voxel_pooling.py:8-72),
backed by the hand-written gfx950 kernel behind `tt_voxel_pool_fwd`.

Same argument meaning, same return layout ([B, C, Y, X] view of a channel-last
buffer), same error behaviour (asserts on non-contiguous inputs; device tensors
required).  Differences that do not change results:
  * no 514 MB `zeros_like(input_features)` is allocated in inference
    (reference: voxel_pooling.py:29) -- the gradient buffer is created in backward;
  * `voxel_num` entries are read once on the host (the reference converts three
    0-dim tensors through pybind on every call: voxel_pooling.py:45-47).
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib


def _pool_launch(B, Np, C, vx, vy, vz, geom_xyz, input_features, out, pos_memo):
    """tt_voxel_pool_fwd_ws with a torch-allocated workspace (atomics-free two-phase kernel); the
    library itself falls back to the single-pass atomic kernel for shapes the fast path does not cover."""
    L = _lib.lib()
    ws_bytes = int(L.tt_voxel_pool_workspace_bytes(ctypes.c_int(B), ctypes.c_int(Np), ctypes.c_int(C),
                                                   ctypes.c_int(vx), ctypes.c_int(vy)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=input_features.device) if ws_bytes > 0 else None
    rc = L.tt_voxel_pool_fwd_ws(
        ctypes.c_int(B), ctypes.c_int(Np), ctypes.c_int(C), ctypes.c_int(vx), ctypes.c_int(vy), ctypes.c_int(vz),
        _lib.ptr(geom_xyz), _lib.ptr(input_features), _lib.ptr(out), _lib.ptr(pos_memo), _lib.ptr(ws),
        ctypes.c_longlong(ws_bytes), _lib.cur_stream(input_features.device))
    _lib.check(rc, "tt_voxel_pool_fwd_ws")


def _voxel_num_tuple(voxel_num):
    if isinstance(voxel_num, torch.Tensor):
        voxel_num = voxel_num.detach().cpu().tolist()
    vx, vy, vz = (int(v) for v in voxel_num)
    return vx, vy, vz


class VoxelPooling(Function):
    @staticmethod
    def forward(ctx, geom_xyz, input_features, voxel_num):
        assert geom_xyz.is_contiguous()
        assert input_features.is_contiguous()
        _lib.require_cuda(geom_xyz, input_features)
        if geom_xyz.dtype != torch.int32:
            raise _lib.TTError("geom_xyz must be int32 (reference: lss.py:630-631 `.int()`)")
        if input_features.dtype != torch.float32:
            raise _lib.TTError("input_features must be float32")
        ctx.mark_non_differentiable(geom_xyz)
        in_shape = input_features.shape
        geom_xyz = geom_xyz.reshape(geom_xyz.shape[0], -1, geom_xyz.shape[-1])
        input_features = input_features.reshape(geom_xyz.shape[0], -1, input_features.shape[-1])
        assert geom_xyz.shape[1] == input_features.shape[1]
        B, Np, C = input_features.shape
        vx, vy, vz = _voxel_num_tuple(voxel_num)
        out = input_features.new_zeros(B, vy, vx, C)
        need_memo = input_features.requires_grad
        pos_memo = geom_xyz.new_full((B, Np, 3), -1) if need_memo else None
        _pool_launch(B, Np, C, vx, vy, vz, geom_xyz, input_features, out, pos_memo)
        if need_memo:
            ctx.save_for_backward(pos_memo)
        ctx.in_shape = in_shape
        ctx.vxy = (vx, vy)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_output_features):
        (pos_memo,) = ctx.saved_tensors
        B, Np, _ = pos_memo.shape
        C = grad_output_features.shape[1]
        vx, vy = ctx.vxy
        g = grad_output_features.permute(0, 2, 3, 1).contiguous()  # [B,Y,X,C]
        grad_in = g.new_empty(B, Np, C)
        rc = _lib.lib().tt_voxel_pool_bwd(
            ctypes.c_int(B), ctypes.c_int(Np), ctypes.c_int(C), ctypes.c_int(vx), ctypes.c_int(vy),
            _lib.ptr(pos_memo), _lib.ptr(g), _lib.ptr(grad_in), _lib.cur_stream(g.device))
        _lib.check(rc, "tt_voxel_pool_bwd")
        return None, grad_in.reshape(ctx.in_shape), None


def voxel_pooling(geom_xyz, input_features, voxel_num):
    return VoxelPooling.apply(geom_xyz, input_features, voxel_num)


def voxel_pooling_forward_wrapper(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y,
                                  num_voxel_z, geom_xyz, input_features, output_features, pos_memo):
    """Argument-for-argument mirror of the reference extension symbol
    (ops/voxel_pooling/src/voxel_pooling_forward.cpp:24-37); returns 1."""
    _lib.require_cuda(geom_xyz, input_features, output_features, pos_memo)
    for name, t in (("geom_xyz", geom_xyz), ("input_features", input_features)):
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous ")
    _pool_launch(int(batch_size), int(num_points), int(num_channels), int(num_voxel_x), int(num_voxel_y),
                 int(num_voxel_z), geom_xyz, input_features, output_features, pos_memo)
    return 1


class VoxelPoolPlan:
    """Static-geometry plan of the voxel-pool forward (tt_voxel_pool_plan_build / tt_voxel_pool_fwd_planned): for a
    fixed camera rig `geom_xyz` is the same every frame (LSS.get_geometry depends on the calibration only), so the
    in-range points are sorted by (sample, cell) ONCE and every forward streams the point rows cell by cell -- no index
    work, no atomics, deterministic.  `plan(feats)` == `voxel_pooling(geom_xyz, feats, voxel_num)`."""

    def __init__(self, geom_xyz, voxel_num):
        _lib.require_cuda(geom_xyz)
        assert geom_xyz.is_contiguous() and geom_xyz.dtype == torch.int32
        g = geom_xyz.reshape(geom_xyz.shape[0], -1, 3)
        self.B, self.Np = g.shape[0], g.shape[1]
        self.vx, self.vy, self.vz = _voxel_num_tuple(voxel_num)
        L, ci = _lib.lib(), ctypes.c_int
        L.tt_voxel_pool_plan_bytes.restype = ctypes.c_longlong
        L.tt_voxel_pool_plan_workspace_bytes.restype = ctypes.c_longlong
        L.tt_voxel_pool_planned_workspace_bytes.restype = ctypes.c_longlong
        pb = int(L.tt_voxel_pool_plan_bytes(ci(self.B), ci(self.Np), ci(self.vx), ci(self.vy)))
        wb = int(L.tt_voxel_pool_plan_workspace_bytes(ci(self.B), ci(self.Np)))
        self.plan = torch.empty(pb, dtype=torch.uint8, device=g.device)
        ws = torch.empty(wb, dtype=torch.uint8, device=g.device)
        _lib.check(L.tt_voxel_pool_plan_build(ci(self.B), ci(self.Np), ci(self.vx), ci(self.vy), ci(self.vz), _lib.ptr(g),
                                              _lib.ptr(ws), ctypes.c_longlong(wb), _lib.ptr(self.plan),
                                              ctypes.c_longlong(pb), _lib.cur_stream(g.device)), "tt_voxel_pool_plan_build")
        self._ws = None

    def forward_into(self, input_features, out):
        """Accumulate into `out` (B, Y, X, C) f32, like voxel_pooling_forward_wrapper does into output_features."""
        _lib.require_cuda(input_features, out)
        f = input_features.reshape(self.B, self.Np, -1)
        assert f.is_contiguous() and f.dtype == torch.float32 and out.is_contiguous()
        C = f.shape[-1]
        L, ci = _lib.lib(), ctypes.c_int
        wb = int(L.tt_voxel_pool_planned_workspace_bytes(ci(self.B), ci(self.Np), ci(C), ci(self.vx), ci(self.vy)))
        if self._ws is None or self._ws.numel() < wb:
            self._ws = torch.empty(wb, dtype=torch.uint8, device=f.device)
        _lib.check(L.tt_voxel_pool_fwd_planned(ci(self.B), ci(self.Np), ci(C), ci(self.vx), ci(self.vy), _lib.ptr(self.plan),
                                               _lib.ptr(f), _lib.ptr(out), _lib.ptr(self._ws), ctypes.c_longlong(wb),
                                               _lib.cur_stream(f.device)), "tt_voxel_pool_fwd_planned")
        return out

    def __call__(self, input_features):
        C = input_features.shape[-1]
        out = input_features.new_zeros(self.B, self.vy, self.vx, C)
        return self.forward_into(input_features, out).permute(0, 3, 1, 2)
