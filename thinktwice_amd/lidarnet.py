"""`LidarNet` -- host-side mirror of the reference LiDAR backbone
(open_loop_training/code/model_code/backbones/lidarnet.py:61-96 + `SparseEncoder_fp32` :24-58; the layer
bodies live in mmdet3d / mmcv / spconv, config at configs/thinktwice.py:159-193).

points (B,Np,5) -> hard voxelise + mean VFE -> sparse 3-D conv encoder -> dense (B,84,84,256)
-> SECOND -> SECONDFPN -> (B,84,84,512), channel-last f32.  The sparse part keeps its active-row
counts on the device (no `coors[-1,0]+1` host sync, lidarnet.py:90).
"""
import ctypes
import os

import torch

from . import _lib, ops, weights
from .layers import conv_from_sd, deconv2x2_from_sd
from .ops import _c, _ll, check, lib, ptr
from .registry import BACKBONES, MIDDLE_ENCODERS

F32 = torch.float32
# A/B knob, default OFF: mask-sorted tiles of the gathered GEMM (tt_sp_tile_plan).  Measured on the bench workload
# (gpurun_out/r2_bench_x3_plan{0,1}.json -> profiles/r02_sparse_tile_plan_ab.txt): it wins only where rows carry ~1 tap
# (first SubM level / first strided conv: 0.87 -> 0.41 ms) and LOSES at 6-23 taps per row (9.5 -> 15.0 ms on the
# 1.65 M-row level): sorting by mask scatters spatial neighbours over different tiles, so the gathered input rows stop
# hitting L2, and the tap union of a tile is nearly full there anyway.
_TILE_PLAN = os.environ.get("TT_SPARSE_TILE_PLAN", "0") == "1"


def _pow2(n):
    p = 1
    while p < n:
        p *= 2
    return p


class _Level:
    """Active sites of one resolution level: coords, device row count, dense index volume, SubM rulebook."""

    def __init__(self, coords, rows, max_rows, dims, batch, dev, vol=None):
        self.coords, self.rows, self.max_rows, self.dims, self.batch = coords, rows, max_rows, list(dims), batch
        self.dims_c = (ctypes.c_int * 3)(*self.dims)
        self.vol = vol
        self._subm = None
        self._subm_plan = None

    def volume(self):
        if self.vol is None:
            D, H, W = self.dims
            self.vol = torch.empty(self.batch * D * H * W, dtype=torch.int32, device=self.coords.device)
            check(lib().tt_sp_volume_build(ptr(self.coords), ptr(self.rows), _ll(self.max_rows), _c(self.batch),
                                           self.dims_c, ptr(self.vol), ops.cur_stream(self.coords.device)),
                  "tt_sp_volume_build")
        return self.vol

    def subm_rulebook(self):
        if self._subm is None:
            g = (ctypes.c_int * 9)(3, 3, 3, 1, 1, 1, 1, 1, 1)
            nbr = torch.empty(self.max_rows, 27, dtype=torch.int32, device=self.coords.device)
            check(lib().tt_sp_rulebook(ptr(self.coords), ptr(self.rows), _ll(self.max_rows), g, self.dims_c,
                                       ptr(self.volume()), ptr(nbr), ops.cur_stream(nbr.device)), "tt_sp_rulebook")
            self._subm = nbr
            self._subm_plan = ops.sp_tile_plan(nbr, self.rows) if _TILE_PLAN else None
        return self._subm

    def subm_plan(self):
        self.subm_rulebook()
        return self._subm_plan


def _sp_weight(w, dev, dtype):
    """spconv (Cout,kD,kH,kW,Cin) -> [Cout][1][KV][Cin_p] (K-contiguous rows of the gathered GEMM)."""
    co, kd, kh, kw, ci = w.shape
    sdt = weights.storage_dtype(dtype)
    vec = weights.vec_of(dtype)
    cp = (ci + vec - 1) // vec * vec
    out = torch.zeros(co, 1, kd * kh * kw, cp, dtype=sdt, device=dev)
    out[..., :ci] = w.to(dev).reshape(co, 1, kd * kh * kw, ci).to(sdt)
    out = out.contiguous()
    # precision mode "f32x3": (plain f32 weights, the same pre-split into bf16 pairs)
    return (out, weights.split_pairs_x3(out) if dtype == weights.X3 else None)


def _bn1d(sd, p, dev, eps=1e-3, momentum=0.01):
    """(folded eval scale, folded eval shift, train-mode spec) of a BatchNorm1d over sparse rows (mmdet3d SparseEncoder
    norm_cfg: eps 1e-3, momentum 0.01)."""
    s = sd[p + ".weight"].float() / torch.sqrt(sd[p + ".running_var"].float() + eps)
    t = sd[p + ".bias"].float() - sd[p + ".running_mean"].float() * s
    from . import layers
    return s.to(dev).contiguous(), t.to(dev).contiguous(), layers._bn_spec(sd, p, dev, eps, momentum)


def _sp_conv(feats, nbr, level, w, bn, res=None, relu=True, plan=None, in_level=None, stride=1):
    """SubMConv3d / SparseConv3d + BN1d (+ residual) + ReLU as ONE gathered MFMA GEMM (with the rulebook's tile plan:
    rows sorted by tap mask, each 256-row tile multiplies only the taps that exist in it).  `in_level`: the input level of
    a strided layer (the training tape transposes its rulebook)."""
    from . import layers
    in_rows = None if in_level is None else (in_level.rows, in_level.max_rows)
    act = _lib.ACT_RELU if relu else _lib.ACT_NONE
    if layers.BN_TRAIN:     # model.train(): raw rulebook GEMM, then batch statistics over the live rows
        z = ops.gather_conv(feats, nbr, level.rows, w[0], w_x3=w[1], plan=plan, in_rows=in_rows, stride=stride, bn_raw=True)
        return ops.batchnorm_train(z, bn[2], act, res1=res, m_dev=level.rows)
    return ops.gather_conv(feats, nbr, level.rows, w[0], scale=bn[0], shift=bn[1], res=res, act=act, w_x3=w[1], plan=plan,
                           in_rows=in_rows, stride=stride)


@MIDDLE_ENCODERS.register_module()
class SparseEncoder_fp32:
    """mmdet3d SparseEncoder body on gfx950 (lidarnet.py:24-58)."""

    def __init__(self, in_channels, sparse_shape, output_channels=128, base_channels=16,
                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), device="cuda",
                 dtype=torch.float32, **kw):
        self.wdtype = dtype
        self.dtype = weights.storage_dtype(dtype)
        self.in_channels = in_channels
        self.sparse_shape = list(sparse_shape)
        self.encoder_channels, self.encoder_paddings = encoder_channels, encoder_paddings
        self.device = torch.device(device)

    def load_state_dict(self, sd, p):
        dev = self.device
        def _sp_w(t, d):
            return _sp_weight(t, d, self.wdtype)

        def _layer(conv, bnn):
            """prepared weight + folded BN of one sparse layer, registered for the training tape"""
            wt = sd[conv + ".weight"]
            w = _sp_w(wt, dev)
            b = _bn1d(sd, bnn, dev)
            from . import autodiff
            autodiff.CONV_META[w[0]] = autodiff.ConvMeta(
                conv, wt.shape[-1], bnn, sd[bnn + ".running_mean"].float().to(dev),
                torch.sqrt(sd[bnn + ".running_var"].float().to(dev) + 1e-3), None, kind="spconv", full_shape=tuple(wt.shape))
            return w, b
        self.w_in, self.bn_in = _layer(p + ".conv_input.0", p + ".conv_input.1")
        self.stages = []
        n = len(self.encoder_channels)
        for i, blocks in enumerate(self.encoder_channels):
            st = []
            for j in range(len(blocks)):
                r = f"{p}.encoder_layers.encoder_layer{i + 1}.{j}"
                if j == len(blocks) - 1 and i != n - 1:
                    pd = self.encoder_paddings[i][j]
                    pd = [pd] * 3 if isinstance(pd, int) else list(pd)
                    st.append(("down", *_layer(r + ".0", r + ".1"), pd))
                else:
                    st.append(("block", *_layer(r + ".conv1", r + ".bn1"), *_layer(r + ".conv2", r + ".bn2")))
            self.stages.append(st)
        self.w_out, self.bn_out = _layer(p + ".conv_out.0", p + ".conv_out.1")
        return self

    def _down(self, feats, lvl, w, bn, kernel, stride, pad, batch):
        dev = feats.device
        od = [(lvl.dims[d] + 2 * pad[d] - kernel[d]) // stride[d] + 1 for d in range(3)]
        cells = batch * od[0] * od[1] * od[2]
        fan = 1
        for d in range(3):
            fan *= -(-kernel[d] // stride[d])          # outputs one input can reach per dim
        max_out = min(lvl.max_rows * fan, cells)
        coords = torch.empty(max_out, 4, dtype=torch.int32, device=dev)
        rows = torch.empty(1, dtype=torch.int32, device=dev)
        vol = torch.empty(cells, dtype=torch.int32, device=dev)
        g = (ctypes.c_int * 9)(*kernel, *stride, *pad)
        odc = (ctypes.c_int * 3)(*od)
        st = ops.cur_stream(dev)
        ws_bytes = int(lib().tt_sp_strided_outputs_workspace_bytes(_ll(cells)))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(lib().tt_sp_strided_outputs(ptr(lvl.coords), ptr(lvl.rows), _ll(lvl.max_rows), _c(batch), g, odc,
                                          ptr(ws), _ll(ws_bytes), ptr(vol), ptr(coords), ptr(rows), _ll(max_out), st),
              "tt_sp_strided_outputs")
        new = _Level(coords, rows, max_out, od, batch, dev, vol=vol)
        KV = kernel[0] * kernel[1] * kernel[2]
        nbr = torch.empty(max_out, KV, dtype=torch.int32, device=dev)
        check(lib().tt_sp_rulebook(ptr(coords), ptr(rows), _ll(max_out), g, lvl.dims_c, ptr(lvl.volume()), ptr(nbr), st),
              "tt_sp_rulebook")
        plan = ops.sp_tile_plan(nbr, rows) if _TILE_PLAN else None
        return _sp_conv(feats, nbr, new, w, bn, plan=plan, in_level=lvl, stride=max(stride)), new

    def forward(self, voxel_features, coors, num_rows, max_rows, batch_size):
        """-> dense channel-last (B, H, W, C*D) f32 (== spatial_features.view(N, C*D, H, W))."""
        dev = voxel_features.device
        lvl = _Level(coors, num_rows, max_rows, self.sparse_shape, batch_size, dev)
        cp = self.w_in[0].shape[-1]
        f0 = torch.zeros(max_rows, cp, dtype=self.dtype, device=dev)      # channel-padded voxel features
        nf = voxel_features.shape[1]
        ops.copy_nhwc(voxel_features.view(max_rows, 1, 1, nf), f0.view(max_rows, 1, 1, cp), C=nf)
        x = _sp_conv(f0, lvl.subm_rulebook(), lvl, self.w_in, self.bn_in, plan=lvl.subm_plan())
        for st in self.stages:
            for item in st:
                if item[0] == "block":
                    _, w1, b1, w2, b2 = item
                    nbr = lvl.subm_rulebook()
                    y = _sp_conv(x, nbr, lvl, w1, b1, plan=lvl.subm_plan())
                    x = _sp_conv(y, nbr, lvl, w2, b2, res=x, plan=lvl.subm_plan())   # relu(bn2(conv2) + identity)
                else:
                    _, w, bn, pd = item
                    x, lvl = self._down(x, lvl, w, bn, (3, 3, 3), (2, 2, 2), pd, batch_size)
        x, lvl = self._down(x, lvl, self.w_out, self.bn_out, (3, 1, 1), (2, 1, 1), (0, 0, 0), batch_size)
        return ops.sp_to_dense(x, lvl.coords, lvl.rows, lvl.max_rows, lvl.dims, batch_size)


@BACKBONES.register_module()
class LidarNet:
    def __init__(self, bev_h=None, bev_w=None, pts_voxel_layer=None, pts_voxel_encoder=None,
                 pts_middle_encoder=None, pts_fusion_layer=None, pts_backbone=None, pts_neck=None,
                 pts_bbox_head=None, train_cfg=None, test_cfg=None, device="cuda", dtype=torch.float32, **kw):
        self.vl, self.bb, self.nk = dict(pts_voxel_layer), dict(pts_backbone), dict(pts_neck)
        me = dict(pts_middle_encoder)
        me.pop("type", None)
        self.wdtype = dtype
        self.dtype = weights.storage_dtype(dtype)
        self.middle = SparseEncoder_fp32(**me, device=device, dtype=dtype)
        self.device = torch.device(device)
        self.training = False

    def load_state_dict(self, sd, prefix="lidar_encoder"):
        dev, p = self.device, prefix
        self.middle.load_state_dict(sd, p + ".pts_middle_encoder")
        eps = self.bb.get("bn_eps", 1e-3)
        mom = 0.01                                   # norm_cfg momentum of SECOND / SECONDFPN (configs/thinktwice.py:183,190)
        self.blocks = []
        for b, (n, s) in enumerate(zip(self.bb["layer_nums"], self.bb["layer_strides"])):
            q = f"{p}.pts_backbone.blocks.{b}"
            self.blocks.append([conv_from_sd(sd, f"{q}.{3 * l}", self.wdtype, dev, bn=f"{q}.{3 * l + 1}", eps=eps,
                                             stride=s if l == 0 else 1, pad=1, act="relu", bn_momentum=mom) for l in range(n + 1)])
        q = p + ".pts_neck.deblocks"
        self.de0 = conv_from_sd(sd, q + ".0.0", self.wdtype, dev, bn=q + ".0.1", eps=eps, act="relu", bn_momentum=mom)
        self.de1 = deconv2x2_from_sd(sd, q + ".1.0", self.wdtype, dev, bn=q + ".1.1", eps=eps, act="relu", bn_momentum=mom)
        return self

    def voxelize(self, pts, max_voxels=None):
        """MVXTwoStageDetector.voxelize + HardSimpleVFE on the device.  mmcv's hard voxelization keeps at most max_voxels
        (120000 train / 160000 eval, per sample) voxels in first-appearance order; a sample has at most one voxel per point,
        so the cap cannot bind while Np <= max_voxels (65536 points per sweep in the reference data): the sorted pipeline
        then runs as is; larger clouds take `tt_lidar_voxelize_capped` (first-appearance ranking of the voxels).
        `max_voxels`: override of the configured cap (tests)."""
        B, Np, nf = pts.shape
        vl = self.vl
        if max_voxels is None:
            cap = vl.get("max_voxels", (120000, 160000))
            max_voxels = (cap[0] if self.training else cap[1]) if isinstance(cap, (tuple, list)) else cap
        rng, vs = vl["point_cloud_range"], vl["voxel_size"]
        grid = [int(round((rng[3 + d] - rng[d]) / vs[d])) for d in range(3)]
        n = B * Np
        L = lib()
        capped = Np > max_voxels
        if capped:
            L.tt_lidar_voxelize_capped_workspace_bytes.restype = ctypes.c_longlong
            ws_bytes = int(L.tt_lidar_voxelize_capped_workspace_bytes(_ll(n)))
        else:
            ws_bytes = int(L.tt_lidar_voxelize_workspace_bytes(_ll(n)))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pts.device)
        feats = torch.empty(n, nf, dtype=F32, device=pts.device)
        coords = torch.empty(n, 4, dtype=torch.int32, device=pts.device)
        num = torch.empty(1, dtype=torch.int32, device=pts.device)
        lo = (ctypes.c_float * 3)(*rng[:3])
        vsz = (ctypes.c_float * 3)(*vs)
        g = (ctypes.c_int * 3)(*grid)
        if capped:
            check(L.tt_lidar_voxelize_capped(ptr(pts), _c(B), _c(Np), _c(nf), lo, vsz, g, _c(self.middle.sparse_shape[0]),
                                             _c(vl["max_num_points"]), _c(int(max_voxels)), ptr(ws), _ll(ws_bytes), ptr(feats),
                                             ptr(coords), ptr(num), ops.cur_stream(pts.device)), "tt_lidar_voxelize_capped")
        else:
            check(L.tt_lidar_voxelize(ptr(pts), _c(B), _c(Np), _c(nf), lo, vsz, g, _c(self.middle.sparse_shape[0]),
                                      _c(vl["max_num_points"]), ptr(ws), _ll(ws_bytes), ptr(feats), ptr(coords),
                                      ptr(num), ops.cur_stream(pts.device)), "tt_lidar_voxelize")
        return feats, coords, num, n

    def forward(self, pts, channel_last=False, rot_flip=False):
        """pts (B,Np,5) f32 device -> [(B,512,84,84)] like the reference (NCHW f32), or the channel-last
        map (optionally with the rot90(flip) of encoder_decoder_framework.py:245-246 applied)."""
        _lib.require_cuda(pts)
        pts = pts.contiguous().float()
        feats, coords, num, max_rows = self.voxelize(pts)
        B = pts.shape[0]
        x = self.middle.forward(feats, coords, num, max_rows, B)          # (B,84,84,256)
        outs = []
        for blk in self.blocks:
            for cv in blk:
                x = cv(x)
            outs.append(x)
        H, W = outs[0].shape[1:3]
        cat = torch.empty(B, H, W, 512, dtype=self.dtype, device=pts.device)
        self.de0(outs[0], out=cat, out_coff=0)
        self.de1(outs[1], out=cat, out_coff=256)
        if rot_flip:
            o = torch.empty(cat.shape, dtype=F32, device=cat.device)     # fusion neck runs in f32
            ops.copy_nhwc(cat, o, rot_flip=True)
            cat = o
        if channel_last:
            return cat
        return [ops.nhwc_to_nchw(cat)]

    __call__ = forward
