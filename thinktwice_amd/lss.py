"""`LSS` camera encoder -- host-side mirror of the reference module
(open_loop_training/code/model_code/backbones/lss.py:351-724), every tensor op on hand-written
gfx950 kernels behind the C ABI.

Same constructor arguments, same state_dict key names, same `forward(img, img_metas)` return
dict (`bev`, `seg`, `fpn_feats`, `lidar2img`, `ida_mat`[, `depth`]) in the reference's NCHW f32
layout.  Internally activations are channel-last; both sweeps go through the image trunk as ONE
batch of B*T*4 images (BN is in eval mode, so per-image results are unchanged), concatenations are
channel-offset writes, the depth softmax / outer product / permute / voxel pooling chain
(lss.py:583-632) is one fused lift-splat kernel that never materialises the 514 MB/sample volume.
"""
import os

import torch

from . import _lib, camera, layers, ops, weights
from .layers import conv_from_sd, conv_from_weight, deconv2x2_from_sd, linear_from_sd, rows, unrows
from .registry import BACKBONES, NECKS, build_neck


class _ResNet50:
    """[3P] mmdet ResNet(depth=50, out_indices 0-3) with eval BN folded into the conv epilogues."""

    def __init__(self, sd, p, dtype, device):
        vec = weights.vec_of(dtype)
        self.cin_pad = vec
        self.stem = conv_from_sd(sd, p + ".conv1", dtype, device, bn=p + ".bn1", stride=2, pad=3, act="relu",
                                 cin_pad=vec)
        # 16-bit modes: "row-run" form of the 7x7/2 stem.  With channel-last pixels of 8 bf16 / f16 (16 B), the 7 taps of one filter
        # row are 112 contiguous bytes: read them as ONE 128 B run of 8 pixels (the 8th meets zero weights) from a
        # zero-bordered image, i.e. a KH=7, KW=1, Cin=64 convolution over a tensor whose "pixels" overlap (pixel
        # stride 8 elements).  K tiles become whole cache lines and the layer runs on the LDS-DMA kernel instead of
        # the 3-channel im2col path (Cin=8 is below every DMA tile).  Same products, different summation order.
        # bf16x3 mode: the same with 4-channel f32 pixels (16 B): 8 pixels = 32 f32 = one 128 B K tile of the x3 kernel.
        self.stem_rr, self.stem_rr_x3 = None, None
        if weights.storage_dtype(dtype) != torch.float32:     # 16-bit storage (bf16 or IEEE half)
            w = sd[p + ".conv1.weight"].to(device)                      # (64, 3, 7, 7)
            wr = torch.zeros(w.shape[0], 7, 1, 64, dtype=dtype, device=device)
            wr[:, :, 0, :56] = torch.nn.functional.pad(w.permute(0, 2, 3, 1), (0, 5)).reshape(w.shape[0], 7, 56).to(dtype)
            self.stem_rr = wr.contiguous()
        elif dtype == weights.X3 and os.environ.get("TT_X3_STEM_ROWRUN", "1") == "1":
            w = sd[p + ".conv1.weight"].to(device).float()
            wr = torch.zeros(w.shape[0], 7, 1, 32, dtype=torch.float32, device=device)
            wr[:, :, 0, :28] = torch.nn.functional.pad(w.permute(0, 2, 3, 1), (0, 1)).reshape(w.shape[0], 7, 28)
            self.stem_rr = wr.contiguous()
            self.stem_rr_x3 = weights.split_pairs_x3(self.stem_rr)
        self.blocks = []
        for li, nb in enumerate((3, 4, 6, 3), start=1):
            stage = []
            for b in range(nb):
                q = f"{p}.layer{li}.{b}"
                s = 2 if (b == 0 and li > 1) else 1
                blk = {
                    "c1": conv_from_sd(sd, q + ".conv1", dtype, device, bn=q + ".bn1", act="relu"),
                    "c2": conv_from_sd(sd, q + ".conv2", dtype, device, bn=q + ".bn2", stride=s, pad=1, act="relu"),
                    "c3": conv_from_sd(sd, q + ".conv3", dtype, device, bn=q + ".bn3", act="relu"),
                    "ds": conv_from_sd(sd, q + ".downsample.0", dtype, device, bn=q + ".downsample.1", stride=s)
                    if b == 0 else None,
                }
                stage.append(blk)
            self.blocks.append(stage)

    def stem_border(self):
        """(top, left, bottom, right) zero border the row-run stem needs around the image, or None."""
        return (3, 3, 3, 5) if self.stem_rr is not None else None

    def __call__(self, x, bordered=False):
        if bordered:      # x: (NI, H+6, W+8, 8) with the image at (3, 3)
            NI, Hp, Wp, _ = x.shape
            if layers.BN_TRAIN:       # model.train(): raw row-run convolution, then the stem's batch-statistics BatchNorm
                z = ops.conv2d(x, self.stem_rr, stride=2, pad=0, act=0, in_cstride=self.cin_pad,
                               out_hw=((Hp - 6) // 2, (Wp - 8) // 2), w_x3=self.stem_rr_x3, out_dtype=torch.float32)
                y = ops.batchnorm_train(z, self.stem.bn, self.stem.act, groups=layers.BN_GROUPS)
            else:
                y = ops.conv2d(x, self.stem_rr, stride=2, pad=0, scale=self.stem.scale, shift=self.stem.shift,
                               act=self.stem.act, in_cstride=self.cin_pad, out_hw=((Hp - 6) // 2, (Wp - 8) // 2),
                               w_x3=self.stem_rr_x3)
        else:
            y = self.stem(x, stop_grad=True)      # (training tape: the image is a leaf without a gradient)
        x = ops.maxpool3x3s2(y)
        outs = []
        for stage in self.blocks:
            for blk in stage:
                idt = blk["ds"](x) if blk["ds"] is not None else x
                # conv1's output has one reader, the 3 x 3 conv2: produced pre-split (pair format, tt_conv_desc.in_pair) -- the
                # operand split once per element in conv1's epilogue instead of nine times per column tile in conv2's K loop
                c2 = blk["c2"]
                rows2 = x.shape[0] * (x.shape[1] // c2.stride) * (x.shape[2] // c2.stride)
                pr = (c2.w_x3 is not None and not layers.BN_TRAIN and
                      ops.pair_ok(min(rows2, x.shape[0] * x.shape[1] * x.shape[2]), c2.w.shape[-1], c2.w.shape[0]))
                y = c2(blk["c1"](x, out_pair=pr), in_pair=pr)
                x = blk["c3"](y, res1=idt)          # relu(bn3(conv3) + identity)
            outs.append(x)
        return outs


@NECKS.register_module(name="PAFPN")
@NECKS.register_module()
class PAFPN_fp32:
    """mmdet PAFPN / the reference's `PAFPN_fp32` (backbones/lss.py:284-348, built by `build_neck(img_neck_conf)` at
    lss.py:402 from `dict(type='PAFPN', in_channels=[256, 512, 1024, 2048], num_outs=4, out_channels=256)`): lateral 1x1
    convs, top-down nearest-upsample adds, 3x3 fpn convs, bottom-up stride-2 adds, 3x3 pafpn convs.  Same constructor
    arguments and state_dict names (`lateral_convs.i.conv`, `fpn_convs.i.conv`, `downsample_convs.i.conv`,
    `pafpn_convs.i.conv`); channel-last tensors.  `targets`: optional [(buffer, channel offset) | None] per output level --
    the consumer's concat buffer an output is produced in (LSS writes the FPN maps straight into the UNet's inputs)."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 dtype=torch.float32, device="cuda", **unused):
        assert start_level == 0 and end_level in (-1, len(in_channels)) and not add_extra_convs and \
            num_outs == len(in_channels), "PAFPN: only the thinktwice.py form (all levels, no extra convs) is built"
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        # dtype "f32x3h" (weights.X3H): the neck's 3 x 3 convolutions read IEEE-half COPIES of their inputs on the two-MFMA h2 product
        # (csrc/conv_h2.hip); its sums (laterals, top-down, bottom-up), inputs and outputs stay f32 (_call_half)
        self.half = isinstance(dtype, str) and dtype == weights.X3H
        self.wdtype, self.device = (weights.X3 if self.half else dtype), torch.device(device)
        self.loaded = False

    def load_state_dict(self, sd, prefix):
        n, dt, dev, L = prefix, self.wdtype, self.device, self.num_outs
        self.lat = [conv_from_sd(sd, f"{n}.lateral_convs.{i}.conv", dt, dev) for i in range(L)]
        self.fpn = [conv_from_sd(sd, f"{n}.fpn_convs.{i}.conv", dt, dev, pad=1) for i in range(L)]
        self.down = [conv_from_sd(sd, f"{n}.downsample_convs.{i}.conv", dt, dev, stride=2, pad=1) for i in range(L - 1)]
        self.paf = [conv_from_sd(sd, f"{n}.pafpn_convs.{i}.conv", dt, dev, pad=1) for i in range(L - 1)]
        if self.half:
            # the same 3 x 3 layers on the h2 kernel (half input copy x f16 (hi, lo) weights); a level with <= 4096 pixel rows in
            # the batch keeps the bf16x3 forms above (the latency kernels have no half / two-output epilogue)
            assert self.out_channels % 64 == 0, "half-storage PAFPN: out_channels must be a multiple of 64"
            h2 = weights.H2
            self.fpn_h = [conv_from_sd(sd, f"{n}.fpn_convs.{i}.conv", h2, dev, pad=1) for i in range(L)]
            self.down_h = [conv_from_sd(sd, f"{n}.downsample_convs.{i}.conv", h2, dev, stride=2, pad=1) for i in range(L - 1)]
            self.paf_h = [conv_from_sd(sd, f"{n}.pafpn_convs.{i}.conv", h2, dev, pad=1) for i in range(L - 1)]
        self.loaded = True
        return self

    def _call_half(self, inputs, targets):
        """"f32x3h": the neck's SUMS stay in f32 -- laterals, the top-down chain, the bottom-up chain, exactly the tensors of the
        f32 flow below -- and every 3 x 3 convolution reads an IEEE-half COPY of its input on the h2 kernel (two f16 MFMAs per
        product).  One rounding per conv input, none accumulating along the chains (tools/precision_mix_emul.py plan `neckop`:
        waypoint L2 0.62 mm on F14, against 1.08 mm with the chains themselves in half).  The copies cost no extra launch: the
        producing epilogue writes the half tensor as its primary output and the f32 one through tt_conv_desc.out2; a level with
        <= 4096 rows runs the plain bf16x3 layers."""
        L, C, f16, f32 = self.num_outs, self.out_channels, torch.float16, torch.float32
        dev = inputs[0].device
        rows = [t.shape[0] * t.shape[1] * t.shape[2] for t in inputs]
        half = [r > 4096 for r in rows]
        lat32, lat16 = [None] * L, [None] * L
        for i in range(L - 1, -1, -1):
            n, h, w, _ = inputs[i].shape
            up = dict(res1=lat32[i + 1], res1_up=True) if i < L - 1 else {}
            if half[i]:
                lat32[i] = torch.empty(n, h, w, C, dtype=f32, device=dev)
                lat16[i] = self.lat[i](inputs[i], out_dtype=f16, out2=lat32[i], **up)       # half copy + the f32 sum
            else:
                lat32[i] = self.lat[i](inputs[i])
                if i < L - 1:
                    ops.upsample_nearest_add_(lat32[i], lat32[i + 1])
        # fpn convs.  Level 0's output IS the neck's first output (f32, in the consumer's buffer) and feeds downsample_convs.0
        if targets[0] is not None:
            buf0, off0 = targets[0]
        else:
            buf0, off0 = torch.empty(inputs[0].shape[0], inputs[0].shape[1], inputs[0].shape[2], C, dtype=f32, device=dev), 0
        inter32, inter16 = [None] * L, [None] * L
        if half[0]:
            inter16[0] = self.fpn_h[0](lat16[0], out2=buf0, out2_coff=off0)
        else:
            self.fpn[0](lat32[0], out=buf0, out_coff=off0)
        for i in range(1, L):
            inter32[i] = self.fpn_h[i](lat16[i], out_dtype=f32) if half[i] else self.fpn[i](lat32[i])
        # bottom-up: inter[i+1] += down(inter[i]); the sum is only read by 3 x 3 convs, so a half level keeps just its half copy
        for i in range(L - 1):
            if half[i]:
                src = inter16[i]
                if half[i + 1]:
                    inter16[i + 1] = self.down_h[i](src, res1=inter32[i + 1], out_dtype=f16)
                    inter32[i + 1] = None
                else:
                    self.down_h[i](src, res1=inter32[i + 1], out=inter32[i + 1])
            elif i == 0:
                self.down[0](buf0, in_coff=off0, cin=C, res1=inter32[1], out=inter32[1])
            else:
                self.down[i](inter32[i], res1=inter32[i + 1], out=inter32[i + 1])
        outs = [(buf0, off0, C)]
        for i in range(1, L):
            conv, src = (self.paf_h[i - 1], inter16[i]) if half[i] else (self.paf[i - 1], inter32[i])
            if targets[i] is not None:
                buf, off = targets[i]
                conv(src, out=buf, out_coff=off)
            else:
                buf, off = conv(src, out_dtype=f32), 0
            outs.append((buf, off, C))
        return outs

    def __call__(self, inputs, targets=None):
        """inputs: the backbone's maps (channel-last).  Returns [(tensor, channel offset, channels)] per level."""
        L = self.num_outs
        assert len(inputs) == len(self.in_channels)
        targets = list(targets) if targets is not None else [None] * L
        C = self.out_channels
        if self.half:
            from . import autodiff
            if autodiff.TAPE is not None or layers.BN_TRAIN:
                raise _lib.TTError("PAFPN: the half-storage mode ('f32x3h') is an inference mode; train in 'f32x3' or float32")
            return self._call_half(inputs, targets)
        from . import autodiff
        lat = [None] * L
        lat[L - 1] = self.lat[L - 1](inputs[L - 1])
        for i in range(L - 2, -1, -1):
            n, h, w, _ = inputs[i].shape
            if autodiff.TAPE is None and not layers.BN_TRAIN and n * h * w > 4096 and inputs[i].dtype == torch.float32:
                # top-down path inside the lateral conv's epilogue: lat[i] = conv(C_i) + nearest_up(lat[i+1]) -- the same two f32
                # additions per element as the separate kernel (bit-identical), without re-reading and re-writing lat[i]
                lat[i] = self.lat[i](inputs[i], res1=lat[i + 1], res1_up=True)
            else:
                lat[i] = self.lat[i](inputs[i])
                ops.upsample_nearest_add_(lat[i], lat[i + 1])
        # out[0] == inter[0] (pafpn.py: outs = [inter_outs[0]] + ...): produced where its consumer wants it
        if targets[0] is not None:
            buf0, off0 = targets[0]
            self.fpn[0](lat[0], out=buf0, out_coff=off0)
        else:
            buf0, off0 = self.fpn[0](lat[0]), 0
        inter = [None] + [self.fpn[i](lat[i]) for i in range(1, L)]
        if L > 1:
            self.down[0](buf0, in_coff=off0, cin=C, res1=inter[1], out=inter[1])
        for i in range(1, L - 1):
            self.down[i](inter[i], res1=inter[i + 1], out=inter[i + 1])
        outs = [(buf0, off0, C)]
        for i in range(1, L):
            if targets[i] is not None:
                buf, off = targets[i]
                self.paf[i - 1](inter[i], out=buf, out_coff=off)
            else:
                buf, off = self.paf[i - 1](inter[i]), 0
            outs.append((buf, off, C))
        return outs


@BACKBONES.register_module()
class LSS:
    def __init__(self, x_bound, y_bound, z_bound, d_bound, final_dim, downsample_factor, output_channels,
                 img_backbone_conf=None, img_neck_conf=None, depth_net_conf=None, seg_net_conf=None,
                 queue_len=1, fpn_in_channels=(64, 128, 256, 512), dtype=torch.float32, device="cuda"):
        self.grid = camera.VoxelGrid(x_bound, y_bound, z_bound)
        self.d_bound = d_bound
        self.final_dim = tuple(final_dim)
        self.downsample_factor = downsample_factor
        self.output_channels = output_channels
        self.queue_len = queue_len
        self.img_neck_conf = dict(img_neck_conf) if img_neck_conf else dict(type="PAFPN", in_channels=[256, 512, 1024, 2048],
                                                                              num_outs=4, out_channels=256)
        # "f32x3h": bf16x3 with the PAFPN on half storage (weights.X3H); every other layer of this module sees plain bf16x3
        self.half_neck = isinstance(dtype, str) and dtype == weights.X3H
        if self.half_neck:
            dtype = weights.X3
        self.wdtype = dtype                               # precision mode of the conv weights (may be weights.X3)
        self.dtype = weights.storage_dtype(dtype)         # storage type of the activations
        self.device = torch.device(device)
        self.frustum = camera.make_frustum(self.final_dim, downsample_factor, d_bound)
        self.depth_channels = self.frustum.shape[0]
        self.voxel_num = self.grid.voxel_num
        self._frustum_dev = None
        self.loaded = False
        self._xpad = {}

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, prefix="img_encoder"):
        dt, dev, p = self.wdtype, self.device, prefix
        f32 = torch.float32
        self.backbone = _ResNet50(sd, p + ".img_backbone", dt, dev)
        # lss.py:402 `self.img_neck = build_neck(img_neck_conf)`: the NECKS registry ('PAFPN' / 'PAFPN_fp32')
        self.img_neck = build_neck(self.img_neck_conf, dtype=weights.X3H if self.half_neck else dt,
                                   device=dev).load_state_dict(sd, p + ".img_neck")
        self.neck_conv = conv_from_sd(sd, p + ".neck_conv", dt, dev)
        d = p + ".depth_net"
        self.bn22 = layers.bn_affine(sd, d + ".bn", dev)
        self.reduce = conv_from_sd(sd, d + ".reduce_conv.0", dt, dev, bn=d + ".reduce_conv.1", pad=1, act="relu")
        self.se = {}
        for name in ("depth", "context"):
            self.se[name] = [
                linear_from_sd(sd, f"{d}.{name}_mlp.fc1", dev, act="relu", in_pad=24),
                linear_from_sd(sd, f"{d}.{name}_mlp.fc2", dev),
                conv_from_sd(sd, f"{d}.{name}_se.conv_reduce", f32, dev, act="relu"),
                conv_from_sd(sd, f"{d}.{name}_se.conv_expand", f32, dev),
            ]
        self.context_conv = conv_from_sd(sd, d + ".context_conv", dt, dev)
        self.bb = []
        for i in range(3):
            q = f"{d}.depth_conv.{i}"
            self.bb.append((conv_from_sd(sd, q + ".conv1", dt, dev, bn=q + ".bn1", pad=1, act="relu"),
                            conv_from_sd(sd, q + ".conv2", dt, dev, bn=q + ".bn2", pad=1, act="relu")))
        a = d + ".depth_conv.3"
        self.aspp = [conv_from_sd(sd, a + ".aspp1.atrous_conv", dt, dev, bn=a + ".aspp1.bn", act="relu")]
        for i, dil in ((2, 6), (3, 12), (4, 18)):
            self.aspp.append(conv_from_sd(sd, f"{a}.aspp{i}.atrous_conv", dt, dev, bn=f"{a}.aspp{i}.bn", pad=dil,
                                          dil=dil, act="relu"))
        self.aspp_gap = conv_from_sd(sd, a + ".global_avg_pool.1", f32, dev, bn=a + ".global_avg_pool.2", act="relu")
        w1 = sd[a + ".conv1.weight"]
        mid = w1.shape[0]
        self.aspp_out = conv_from_sd(sd, a + ".conv1", dt, dev, bn=a + ".bn1", act="relu",
                                     weight=w1[:, : 4 * mid].contiguous())
        # global-pool branch enters conv1 as a per-image shift: bn_scale * (W[:, 4*mid:] @ x5)
        self.aspp_gapw = conv_from_sd(sd, a + ".conv1", f32, dev, weight=w1[:, 4 * mid:].contiguous(), cin_lo=4 * mid)
        self.aspp_gapw.scale, self.aspp_gapw.shift = self.aspp_out.scale, None
        q = d + ".depth_conv.4"
        self.dcn_off = conv_from_sd(sd, q + ".conv_offset", dt, dev, pad=1)
        wd = sd[q + ".weight"]
        self.dcn_groups = mid // wd.shape[1]
        og = wd.shape[0] // self.dcn_groups
        self.dcn_w = []
        for g in range(self.dcn_groups):
            wg = wd[g * og:(g + 1) * og].to(dev)                        # (og, cg, 3, 3)
            wg = wg.permute(0, 2, 3, 1).reshape(og, 1, 9, wg.shape[1])  # [Cout][1][tap][cin]
            self.dcn_w.append(conv_from_weight(wg.to(self.dtype).contiguous(), dt))
            from . import autodiff
            autodiff.CONV_META[self.dcn_w[-1].w] = autodiff.ConvMeta(q, wd.shape[1], kind="dcn_group",
                                                                         full_shape=tuple(wd.shape), lo=g * og)
        self.depth_out = conv_from_sd(sd, d + ".depth_conv.5", dt, dev)
        s = p + ".seg_net"
        self.up = {k: deconv2x2_from_sd(sd, f"{s}.{k}.up", dt, dev) for k in ("unet_layer4", "unet_layer3", "unet_layer2")}
        self.upc = {k: conv_from_sd(sd, f"{s}.{k}.conv_relu.0", dt, dev, pad=1, act="relu")
                    for k in ("unet_layer4", "unet_layer3", "unet_layer2")}
        self.u0a = conv_from_sd(sd, s + ".unet_layer0.1", dt, dev, pad=1, act="relu")
        self.u0b = conv_from_sd(sd, s + ".unet_layer0.3", dt, dev, pad=1)
        self.seg_cp = 12 if self.dtype == f32 else 16
        self.conv_last = conv_from_sd(sd, s + ".conv_last", dt, dev)
        # Inference: `conv_last` (1 x 1, 64 -> n_class, lss.py:272,281) follows `unet_layer0.3` (3 x 3, 64 -> 64, no bias,
        # lss.py:271) with nothing in between, and the 64-channel map between them has no other reader: the two linear maps are ONE
        # 3 x 3 convolution 64 -> n_class with W = W_last . W_3 (composed in f64), which neither writes nor re-reads the
        # (NI, 224, 448, 64) f32 intermediate (1.6 GB at B = 8).  The training tape keeps the two layers (their own gradients).
        w3 = sd[s + ".unet_layer0.3.weight"].double()
        wl = sd[s + ".conv_last.weight"].double()[:, :, 0, 0]
        wm = torch.einsum("om,mckl->ockl", wl.to(w3.device), w3).float().contiguous()
        bl = sd.get(s + ".conv_last.bias")
        b3 = sd.get(s + ".unet_layer0.3.bias")
        bm = None if (bl is None and b3 is None) else \
            ((0 if bl is None else bl.double()) + (0 if b3 is None else wl.to(b3.device) @ b3.double())).float()
        fused = {"w.weight": wm}
        if bm is not None:
            fused["w.bias"] = bm
        self.seg_fused = conv_from_sd(fused, "w", dt, dev, pad=1)
        r = p + ".seg_res_to_image_feature"
        self.seg2feat = []
        for idx, (st, pd) in zip((0, 3, 6, 9, 12, 15, 18), ((1, 0), (1, 0), (2, 1), (1, 0), (2, 1), (1, 0), (2, 1))):
            self.seg2feat.append(conv_from_sd(sd, f"{r}.{idx}", dt, dev, bn=f"{r}.{idx + 1}", stride=st, pad=pd,
                                              act="relu", cin_pad=self.seg_cp if idx == 0 else None))
        self.merge = conv_from_sd(sd, p + ".merge_seg_and_image", dt, dev, pad=1)
        if self.queue_len != 1:
            self.bev_merge = conv_from_sd(sd, p + ".bev_multiframe_merge", f32, dev, pad=1)
        self._frustum_dev = self.frustum.to(dev)
        self.loaded = True
        return self

    # ------------------------------------------------------------------ forward pieces
    def _trunk(self, x, bordered=False):
        """ResNet-50 + PAFPN + neck_conv on NI channel-last images.  FPN maps are produced directly
        inside the UNet concat buffers (torch.cat of lss.py:256 becomes a channel offset)."""
        NI = x.shape[0]
        c = self.backbone(x, bordered=bordered)
        H0, W0 = c[0].shape[1:3]
        dt, dev = self.dtype, x.device
        cat2 = torch.empty(NI, H0, W0, 384, dtype=dt, device=dev)            # [up(d3) 128 | e1 256]
        cat3 = torch.empty(NI, H0 // 2, W0 // 2, 512, dtype=dt, device=dev)  # [up(d4) 256 | e2 256]
        cat4 = torch.empty(NI, H0 // 4, W0 // 4, 512, dtype=dt, device=dev)  # [up(e4) 256 | e3 256]
        # the PAFPN outputs are produced in place inside the UNet concat buffers
        outs = self.img_neck(c, targets=[(cat2, 128), (cat3, 256), (cat4, 256), None])
        return (cat2, cat3, cat4, outs[3][0])

    def _fpn_views(self, bufs):
        """[(tensor, channel offset, channels)] of the 4 FPN maps."""
        cat2, cat3, cat4, e4 = bufs
        return [(cat2, 128, 256), (cat3, 256, 256), (cat4, 256, 256), (e4, 0, 256)]

    def _se_gate(self, name, m):
        fc1, fc2, cr, ce = self.se[name]
        return unrows(ce(cr(fc2(fc1(rows(m))))))

    def _depth_net(self, src, mlp_in, T):
        """DepthNet.forward (lss.py:205-240): returns depth logits f32 (NI,h,w,D) and the merge
        buffer whose channels [0:256] hold the context."""
        NI, h, w, _ = src.shape
        dev, dt = src.device, self.dtype
        m24 = torch.zeros(mlp_in.shape[0], 24, dtype=torch.float32, device=dev)
        # BatchNorm1d(22) of the camera-parameter vector (train mode: batch statistics).  The reference calls
        # _forward_depth_net once per sweep, every time on the KEY frame's mlp_input (lss.py:689-714), so depth_net.bn's
        # running statistics take T momentum updates per iteration with the same batch statistics
        self.bn22(mlp_in, out=m24, running_updates=T)
        x = self.reduce(src)
        dbg = getattr(self, "_dbg", None)          # tools/debug_trainmode.py: intermediate captures
        if dbg is not None:
            dbg.update(src=src, m24=m24, reduce=x)
        g_ctx = ops.repeat_rows(self._se_gate("context", m24), T)
        g_dep = ops.repeat_rows(self._se_gate("depth", m24), T)
        merge_in = torch.empty(NI, h, w, 384, dtype=dt, device=dev)
        self.context_conv(ops.channel_gate(x, g_ctx), out=merge_in, out_coff=0)
        d = ops.channel_gate(x, g_dep)
        if dbg is not None:
            dbg.update(se_depth=d)
        x3 = self.bb[0][0].w_x3 is not None and not layers.BN_TRAIN
        pr = x3 and ops.pair_ok(NI * h * w, d.shape[-1], d.shape[-1])    # conv1's output: one reader, conv2 (pair format)
        for c1, c2 in self.bb:
            d = c2(c1(d, out_pair=pr), res1=d, in_pair=pr)
        if dbg is not None:
            dbg.update(blocks=d)
        mid = d.shape[-1]
        cat = torch.empty(NI, h, w, 4 * mid, dtype=dt, device=dev)
        pr_cat = x3 and mid % 16 == 0 and ops.pair_ok(NI * h * w, 4 * mid, mid)   # the branch concat: one reader, conv1 of the ASPP
        for i, br in enumerate(self.aspp):
            br(d, out=cat, out_coff=i * mid, out_pair=pr_cat)
        x5 = self.aspp_gap(rows(ops.spatial_pool(d, 0)))                 # (NI,1,1,mid) f32
        # eval: bn_scale * (W5 @ x5) joins the folded epilogue; train: the raw W5 @ x5 joins the raw conv in front of the BN
        self.aspp_gapw.scale = None if layers.BN_TRAIN else self.aspp_out.scale
        shift_n = unrows(self.aspp_gapw(x5)).contiguous()
        d = self.aspp_out(cat, shift_n=shift_n, shift_n_mod=NI, in_pair=pr_cat)
        if dbg is not None:
            dbg.update(aspp_cat=cat, x5=x5, aspp_pre_dropout=d)
        if layers.BN_TRAIN:
            d = ops.dropout(d, 0.5)                                      # ASPP nn.Dropout(0.5), lss.py:91,110
        off = self.dcn_off(d, out_dtype=torch.float32)
        cols = ops.deform_im2col3x3(d, off, pad=1)
        dd = torch.empty(NI, h, w, mid, dtype=dt, device=dev)
        cg = mid // self.dcn_groups
        og = self.dcn_w[0].w.shape[0]
        for g, wg in enumerate(self.dcn_w):
            wg(cols, in_coff=g * cg, cin=cg, out=dd.view(NI * h * w, 1, 1, mid), out_coff=g * og)
        depth = self.depth_out(dd, out_dtype=torch.float32)
        return depth, merge_in

    def _seg_net(self, bufs):
        """UNet.forward (lss.py:275-282) -> seg logits (NI, H/2, W/2, seg_cp)."""
        cat2, cat3, cat4, e4 = bufs
        self.up["unet_layer4"](e4, out=cat4, out_coff=0)
        d4 = self.upc["unet_layer4"](cat4)
        self.up["unet_layer3"](d4, out=cat3, out_coff=0)
        d3 = self.upc["unet_layer3"](cat3)
        self.up["unet_layer2"](d3, out=cat2, out_coff=0)
        d2 = self.upc["unet_layer2"](cat2)
        from . import autodiff
        NI, h2, w2, _ = d2.shape
        H, W = 2 * h2, 2 * w2
        fused = autodiff.TAPE is None and not layers.BN_TRAIN and self.dtype == torch.float32
        # the upsampled map and unet_layer0.1's output have ONE reader each, a bf16x3 convolution: they are produced pre-split
        # (pair format, tt_conv_desc.in_pair) -- same sums, the operand split once per element instead of once per tap
        x3 = self.u0a.w_x3 is not None
        p_up = x3 and ops.pair_ok(NI * H * W, 128, 64)
        p_mid = fused and x3 and self.seg_fused.w_x3 is not None and ops.pair_ok(NI * H * W, 64, self.seg_fused.w.shape[0])
        d1 = self.u0a(ops.bilinear_up2(d2, out_pair=p_up), in_pair=p_up, out_pair=p_mid)
        seg = torch.zeros(NI, H, W, self.seg_cp, dtype=self.dtype, device=d1.device)
        if fused:
            self.seg_fused(d1, out=seg, in_pair=p_mid)   # conv_last o unet_layer0.3 as one convolution (load_state_dict)
        else:
            self.conv_last(self.u0b(d1), out=seg)
        return seg

    def geometry(self, gm, batch_size, num_cams):
        return ops.frustum_voxel_index(self._frustum_dev, gm, self.grid.lower, self.grid.voxel_size.tolist(),
                                       batch_size, num_cams)

    @staticmethod
    def host_constants(img_metas, num_cams=4):
        """Everything the forward derives from `img_metas` on the HOST (small CPU tensors): geometry matrix pairs
        (LSS:502-512 inputs), the camera-aware DepthNet vector (LSS:206-231) and the key frame's lidar2img / ida
        for the look module.  Split out so a captured HIP graph can take them as device-resident inputs."""
        mats = camera.stack_img_metas(img_metas, num_cams)
        return {"gm": camera.geometry_matrices(mats, -1), "mlp_in": camera.depth_mlp_input(mats),
                "lidar2img": mats["lidar2img"], "ida_mat": mats["ida_mat"]}

    # ------------------------------------------------------------------ forward
    def forward(self, img, img_metas, timestamps=None, is_return_depth=False, channel_last=False, consts=None,
                prev_bev=None):
        """img (B,T,N,3,H,W) f32 on device (key frame = last T index) -> dict like LSS.forward.
        `consts`: device copies of `host_constants(img_metas)` (graph replay); derived and uploaded here if None.
        `prev_bev`: (B,vy,vx,C) f32 channel-last splat of the previous sweep taken from an earlier call's
        `outs["_key_bev_cl"]` (closed-loop cache): the reference encodes AND splats the older sweep with
        the key frame's matrices (lss.py:206-210, 606-611, 710-717), so with a fixed rig its BEV equals the key-sweep
        BEV of the tick that image was the key frame; only the key sweep is then run through the camera trunk."""
        if not self.loaded:
            raise _lib.TTError("LSS: load_state_dict() first")
        _lib.require_cuda(img)
        if img.dim() == 5:
            img = img.unsqueeze(1)
        B, T, N, C, H, W = img.shape
        assert T == self.queue_len or (prev_bev is not None and T >= 1), "LSS.queue_len must be set correctly in config!"
        assert (H, W) == self.final_dim
        T_all = self.queue_len
        if prev_bev is not None:
            img = img[:, T - 1:]            # key frame only
            T = 1
        if consts is None:
            consts = {k: v.to(img.device) for k, v in self.host_constants(img_metas, N).items()}
        NI = T * B * N
        # sweep-major image order: key sweep first (index 0 == reference sweep index -1)
        from . import autodiff
        # (the row-run stem has no backward yet: a taped forward uses the plain 7x7 form)
        border = self.backbone.stem_border() if (H % 2 == 0 and W % 2 == 0 and autodiff.TAPE is None) else None
        if border is not None:
            # zero-bordered image buffer for the row-run stem; the border is written once, the interior every call
            # (one per stream: with batches pipelined on alternating streams the next forward's image upload must not overwrite
            # a buffer the previous forward's stem may still be reading)
            key = (NI, H, W, str(img.device), int(torch.cuda.current_stream(img.device).cuda_stream))
            x = self._xpad.get(key)
            if x is None:
                x = torch.zeros(NI, H + border[0] + border[2], W + border[1] + border[3], self.backbone.cin_pad,
                                dtype=self.dtype, device=img.device)
                self._xpad[key] = x      # every bordered buffer stays alive: captured HIP graphs hold raw pointers
        else:
            x = torch.empty(NI, H, W, self.backbone.cin_pad, dtype=self.dtype, device=img.device)
        for s in range(T):
            for b in range(B):
                src = img[b, T - 1 - s].contiguous()
                o = (s * B + b) * N
                if border is not None:
                    ops.nchw_to_nhwc_border(src, x[o:o + N], border[0], border[1])
                else:
                    ops.check(ops.lib().tt_nchw_to_nhwc_pad(ops.ptr(src), ops.ptr(x[o:o + N]), N, C, H, W,
                                                            x.shape[-1], ops.dtype_code(x),
                                                            ops.cur_stream(img.device)), "tt_nchw_to_nhwc_pad")
        # train mode: the T sweeps are T equal image groups; every BatchNorm below normalises each with its own batch
        # statistics, like the reference's one-pass-per-sweep loop (lss.py:690-717)
        with layers.bn_groups(T):
            bufs = self._trunk(x, bordered=border is not None)
            fpn2_buf, fpn2_off, _ = self._fpn_views(bufs)[2]
            src = self.neck_conv(fpn2_buf, in_coff=fpn2_off, cin=256)
            mlp_in = consts["mlp_in"]
            depth, merge_in = self._depth_net(src, mlp_in, T)
            # keep FPN maps of the key sweep before the UNet overwrites nothing of them (offset slices)
            seg = self._seg_net(bufs)
            f = seg
            for i, cv in enumerate(self.seg2feat[:-1]):
                f = cv(f, stop_grad=(i == 0))          # lss.py:589: the seg logits enter this branch detached
            self.seg2feat[-1](f, out=merge_in, out_coff=256)
            ctx = self.merge(merge_in, out_dtype=torch.float32)
        geom = self.geometry(consts["gm"], B, N)
        vx, vy, vz = (int(v) for v in self.voxel_num)
        OC = self.output_channels
        bev_cat = torch.zeros(B, vy, vx, OC * T_all, dtype=torch.float32, device=img.device)
        BN = B * N
        for s in range(T):
            ops.lift_splat(depth[s * BN:(s + 1) * BN], ctx[s * BN:(s + 1) * BN], geom, (vx, vy, vz), B, N,
                           out=bev_cat, out_coff=s * OC, record=(s == 0))     # lss.py:711: older sweeps carry no grad
        if prev_bev is not None:
            assert T_all == 2 and tuple(prev_bev.shape) == (B, vy, vx, OC), prev_bev.shape
            bev_cat[..., OC:2 * OC].copy_(prev_bev)
        bev = self.bev_merge(bev_cat) if T_all > 1 else bev_cat
        fpn = [(t[:BN], off, c) for (t, off, c) in self._fpn_views(bufs)]
        outs = {"lidar2img": consts["lidar2img"], "ida_mat": consts["ida_mat"], "_fpn_cl": fpn, "_bev_cl": bev,
                "_geom": geom, "_key_bev_cl": bev_cat[..., :OC], "_seg_cl": seg[:BN], "_depth_cl": depth[:BN]}
        if channel_last:
            return outs
        outs["bev"] = ops.nhwc_to_nchw(bev)
        outs["seg"] = ops.nhwc_to_nchw(seg[:BN], C=12)
        outs["fpn_feats"] = tuple(ops.nhwc_to_nchw(t, C=c, coff=off) for (t, off, c) in fpn)
        if is_return_depth:
            outs["depth"] = ops.nhwc_to_nchw(depth[:BN])
        return outs

    __call__ = forward
