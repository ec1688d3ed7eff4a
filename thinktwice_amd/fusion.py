"""BEV fusion + flatten network -- mirror of EncoderDecoder.build_fusion_and_flatten_network_for_BEV /
get_fusion_feat (open_loop_training/code/encoder_decoder_framework.py:81-138,213-235) and the
SEBasicBlock / SEModule of code/utils.py:84-121.  f32, channel-last, one kernel launch per layer with
BN / bias / ReLU / residual fused into the conv epilogue.
"""
import torch

from . import _lib, layers, ops
from .layers import conv_from_sd, linear_from_sd, rows, unrows

F32 = torch.float32


class SEBlock:
    """SEBasicBlock (utils.py:99-121): conv-bn-relu x2, SE gate (pool = mean/2 + max/2), +x, relu."""

    def __init__(self, sd, name, dev):
        self.c1 = conv_from_sd(sd, name + ".conv1", F32, dev, bn=name + ".bn1", pad=1, act="relu")
        self.c2 = conv_from_sd(sd, name + ".conv2", F32, dev, bn=name + ".bn2", pad=1, act="relu")
        self.fc1 = conv_from_sd(sd, name + ".se.fc1", F32, dev, act="relu")
        self.fc2 = conv_from_sd(sd, name + ".se.fc2", F32, dev)

    def __call__(self, x):
        y = self.c2(self.c1(x))
        g = unrows(self.fc2(self.fc1(rows(ops.spatial_pool(y, 1)))))
        return ops.channel_gate(y, g, res=x, gate_act=_lib.ACT_SIGMOID, out_act=_lib.ACT_RELU)


class FlattenTail:
    """conv21_10 -> MLP10 -> conv10_4 -> MLP4 -> conv4_2 -> MLP2 -> output_fc (EDF:228-234), shared with
    the decoder's grid2feat (thinktwice_decoder.py:405-415)."""

    def __init__(self, sd, dev):
        self.c21 = conv_from_sd(sd, "conv21_10", F32, dev, stride=2, act="relu")
        self.m10 = SEBlock(sd, "MLP10", dev)
        self.c10 = conv_from_sd(sd, "conv10_4", F32, dev, stride=2, act="relu")
        self.m4 = SEBlock(sd, "MLP4", dev)
        self.c4 = conv_from_sd(sd, "conv4_2", F32, dev, stride=1, act="relu")
        self.m2 = SEBlock(sd, "MLP2", dev)
        # flatten(start_dim=1) of (N,256,2,2) is channel-major; our map is (N,2,2,256): permute fc columns
        w = sd["output_fc.0.weight"]
        wp = w.view(w.shape[0], 256, 4).permute(0, 2, 1).reshape(w.shape[0], 1024).contiguous()
        self.fc0 = linear_from_sd({"w.weight": wp, "w.bias": sd["output_fc.0.bias"]}, "w", dev, act="relu")
        from . import autodiff            # training tape: the gradient goes back under the reference name / column order
        autodiff.CONV_META[self.fc0.w] = autodiff.ConvMeta("output_fc.0", 1024, None, None, None, self.fc0.shift,
                                                               kind="linear_hwc", lo=4)
        self.bn = layers.bn_affine(sd, "output_fc.2", dev)
        self.fc3 = linear_from_sd(sd, "output_fc.3", dev, act="relu")

    def __call__(self, f21, want_mids=False):
        f10 = self.m10(self.c21(f21))
        f4 = self.m4(self.c10(f10))
        f2 = self.m2(self.c4(f4))
        h = unrows(self.fc0(f2.view(f2.shape[0], 1, 1, 1024)))
        h = self.bn(h)                                   # BatchNorm1d (output_fc.2)
        flat = unrows(self.fc3(rows(h)))
        return (flat, [f10, f4, f2]) if want_mids else flat


class BEVFusion:
    def __init__(self, sd, dev):
        def pair(name, stride):
            return (conv_from_sd(sd, name + ".0", F32, dev, bn=name + ".1", stride=stride, pad=1, act="relu"),
                    name, stride)
        self.dev = dev
        self.cam0 = conv_from_sd(sd, "conv_cam.0", F32, dev, bn="conv_cam.1", pad=1, act="relu")
        self.cam3 = conv_from_sd(sd, "conv_cam.3", F32, dev, bn="conv_cam.4", pad=1, act="relu")      # + bev, relu
        self.lid0 = conv_from_sd(sd, "conv_lidar.0", F32, dev, bn="conv_lidar.1", stride=2, pad=1, act="relu")
        self.lid3 = conv_from_sd(sd, "conv_lidar.3", F32, dev, bn="conv_lidar.4", stride=2, pad=1, act="relu")
        self.fus0 = conv_from_sd(sd, "conv_fusion.0", F32, dev, bn="conv_fusion.1", pad=1, act="relu")
        self.fus3 = conv_from_sd(sd, "conv_fusion.3", F32, dev, bn="conv_fusion.4", pad=1, act="relu")  # + cam + pts
        self.to32 = conv_from_sd(sd, "_256_to_32", F32, dev, pad=1, act="relu")
        self.m21 = SEBlock(sd, "MLP21", dev)
        self.tail = FlattenTail(sd, dev)

    def __call__(self, cam_bev, lidar_bev):
        """cam_bev (B,21,21,256) f32, lidar_bev (B,84,84,512) f32 (both already rot90/flipped) ->
        flat (B,256), bev32 (B,21,21,32), mids."""
        B, H, W, _ = cam_bev.shape
        cat = torch.empty(B, H, W, 512, dtype=F32, device=cam_bev.device)
        self.cam3(self.cam0(cam_bev), res1=cam_bev, out=cat, out_coff=0)
        self.lid3(self.lid0(lidar_bev), out=cat, out_coff=256)
        bev = self.fus3(self.fus0(cat), res1=cat, res1_coff=0, res2=cat, res2_coff=256)
        f21 = self.m21(self.to32(bev))
        flat, mids = self.tail(f21, want_mids=True)
        return flat, f21, [None, None, f21] + mids
