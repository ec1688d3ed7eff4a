"""state_dict layout of the reference model + a seeded, name-keyed random initialiser.

`param_spec()` lists every tensor of the reference `EncoderDecoder.state_dict()` (names/shapes of
SURVEY.md Appendix B; in-repo modules are checked against the instantiated reference classes by
tests/golden/gen_golden.py, third-party ones follow the pinned package versions), so an mmcv-style
checkpoint (`load_checkpoint`, AGENT:170) loads unchanged.

`init_params(seed)` fills it deterministically *per key* (generator seeded by crc32(name) ^ seed),
independent of construction order, so the GPU box regenerates bit-identical weights.  The reference
checkpoint is not available offline (README.md:21), so all parity work uses these weights.
"""
import math
import zlib
from collections import OrderedDict

import torch


def _bn(spec, p, c):
    spec[p + ".weight"] = ((c,), "bn_w")
    spec[p + ".bias"] = ((c,), "bn_b")
    spec[p + ".running_mean"] = ((c,), "bn_mean")
    spec[p + ".running_var"] = ((c,), "bn_var")
    spec[p + ".num_batches_tracked"] = ((), "bn_nbt")


def _conv(spec, p, cout, cin, k, bias=True, kind="w"):
    kh, kw = (k, k) if isinstance(k, int) else k
    spec[p + ".weight"] = ((cout, cin, kh, kw), kind)
    if bias:
        spec[p + ".bias"] = ((cout,), "b")


def _lin(spec, p, cout, cin, bias=True, kind="w"):
    spec[p + ".weight"] = ((cout, cin), kind)
    if bias:
        spec[p + ".bias"] = ((cout,), "b")


def _ln(spec, p, c):
    spec[p + ".weight"] = ((c,), "ln_w")
    spec[p + ".bias"] = ((c,), "ln_b")


def _resnet50(spec, p):
    _conv(spec, p + ".conv1", 64, 3, 7, bias=False)
    _bn(spec, p + ".bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
        for b in range(blocks):
            q = f"{p}.layer{li}.{b}"
            _conv(spec, q + ".conv1", planes, inplanes, 1, bias=False)
            _bn(spec, q + ".bn1", planes)
            _conv(spec, q + ".conv2", planes, planes, 3, bias=False)
            _bn(spec, q + ".bn2", planes)
            _conv(spec, q + ".conv3", planes * 4, planes, 1, bias=False)
            _bn(spec, q + ".bn3", planes * 4)
            if b == 0:
                _conv(spec, q + ".downsample.0", planes * 4, inplanes, 1, bias=False)
                _bn(spec, q + ".downsample.1", planes * 4)
            inplanes = planes * 4


def _pafpn(spec, p, in_channels=(256, 512, 1024, 2048), out=256):
    for i, c in enumerate(in_channels):
        _conv(spec, f"{p}.lateral_convs.{i}.conv", out, c, 1)
    for i in range(len(in_channels)):
        _conv(spec, f"{p}.fpn_convs.{i}.conv", out, out, 3)
    for i in range(len(in_channels) - 1):
        _conv(spec, f"{p}.downsample_convs.{i}.conv", out, out, 3)
    for i in range(len(in_channels) - 1):
        _conv(spec, f"{p}.pafpn_convs.{i}.conv", out, out, 3)


def _depth_net(spec, p, cin=512, mid=512, ctx=256, depth=80):
    _conv(spec, p + ".reduce_conv.0", mid, cin, 3)
    _bn(spec, p + ".reduce_conv.1", mid)
    _conv(spec, p + ".context_conv", ctx, mid, 1)
    _bn(spec, p + ".bn", 22)
    for n in ("depth", "context"):
        _lin(spec, f"{p}.{n}_mlp.fc1", mid, 22)
        _lin(spec, f"{p}.{n}_mlp.fc2", mid, mid)
        _conv(spec, f"{p}.{n}_se.conv_reduce", mid, mid, 1)
        _conv(spec, f"{p}.{n}_se.conv_expand", mid, mid, 1)
    for i in range(3):  # mmdet BasicBlock x3
        q = f"{p}.depth_conv.{i}"
        _conv(spec, q + ".conv1", mid, mid, 3, bias=False)
        _bn(spec, q + ".bn1", mid)
        _conv(spec, q + ".conv2", mid, mid, 3, bias=False)
        _bn(spec, q + ".bn2", mid)
    q = f"{p}.depth_conv.3"  # ASPP
    _conv(spec, q + ".aspp1.atrous_conv", mid, mid, 1, bias=False)
    _bn(spec, q + ".aspp1.bn", mid)
    for i in (2, 3, 4):
        _conv(spec, f"{q}.aspp{i}.atrous_conv", mid, mid, 3, bias=False)
        _bn(spec, f"{q}.aspp{i}.bn", mid)
    _conv(spec, q + ".global_avg_pool.1", mid, mid, 1, bias=False)
    _bn(spec, q + ".global_avg_pool.2", mid)
    _conv(spec, q + ".conv1", mid, mid * 5, 1, bias=False)
    _bn(spec, q + ".bn1", mid)
    q = f"{p}.depth_conv.4"  # mmcv DCN (DeformConv2dPack, groups 4, no bias)
    spec[q + ".weight"] = ((mid, mid // 4, 3, 3), "w")
    _conv(spec, q + ".conv_offset", 18, mid, 3, kind="dcn_off_w")
    spec[q + ".conv_offset.bias"] = ((18,), "dcn_off_b")
    _conv(spec, f"{p}.depth_conv.5", depth, mid, 1)


def _lss(spec, p, cfg):
    enc = cfg["img_encoder"]
    H, W = enc["final_dim"]
    D = int(round((enc["d_bound"][1] - enc["d_bound"][0]) / enc["d_bound"][2]))
    ds = enc["downsample_factor"]
    spec[p + ".voxel_size"] = ((3,), "buf_voxel_size")
    spec[p + ".voxel_coord"] = ((3,), "buf_voxel_coord")
    spec[p + ".voxel_num"] = ((3,), "buf_voxel_num")
    spec[p + ".frustum"] = ((D, H // ds, W // ds, 4), "buf_frustum")
    _conv(spec, p + ".bev_multiframe_merge", 256, 256 * enc["queue_len"], 3, bias=False)
    _resnet50(spec, p + ".img_backbone")
    _pafpn(spec, p + ".img_neck")
    _conv(spec, p + ".neck_conv", 512, 256, 1)
    _depth_net(spec, p + ".depth_net", depth=D)
    s = p + ".seg_net"
    for name, cin, cmid, cout in (("unet_layer4", 256, 512, 256), ("unet_layer3", 256, 512, 256),
                                  ("unet_layer2", 256, 384, 128)):
        spec[f"{s}.{name}.up.weight"] = ((cin, cout, 2, 2), "w_deconv")
        spec[f"{s}.{name}.up.bias"] = ((cout,), "b")
        _conv(spec, f"{s}.{name}.conv_relu.0", cout, cmid, 3)
    _conv(spec, s + ".unet_layer0.1", 64, 128, 3, bias=False)
    _conv(spec, s + ".unet_layer0.3", 64, 64, 3, bias=False)
    _conv(spec, s + ".conv_last", 12, 64, 1)
    q = p + ".seg_res_to_image_feature"
    for idx, (co, ci, k) in zip((0, 3, 6, 9, 12, 15, 18),
                                ((64, 12, 1), (16, 64, 1), (32, 16, 3), (32, 32, 1), (64, 32, 3),
                                 (64, 64, 1), (128, 64, 3))):
        _conv(spec, f"{q}.{idx}", co, ci, k)
        _bn(spec, f"{q}.{idx + 1}", co)
    _conv(spec, p + ".merge_seg_and_image", 256, 384, 3)


def _lidar(spec, p, cfg):
    me = cfg["lidar_encoder"]["pts_middle_encoder"]
    q = p + ".pts_middle_encoder"
    base = me["base_channels"]
    # spconv v2 weight layout (Cout, kD, kH, kW, Cin)
    spec[q + ".conv_input.0.weight"] = ((base, 3, 3, 3, me["in_channels"]), "w_sp")
    _bn(spec, q + ".conv_input.1", base)
    cin = base
    chans = me["encoder_channels"]
    for i, blocks in enumerate(chans):
        for j, cout in enumerate(blocks):
            r = f"{q}.encoder_layers.encoder_layer{i + 1}.{j}"
            if j == len(blocks) - 1 and i != len(chans) - 1:
                spec[r + ".0.weight"] = ((cout, 3, 3, 3, cin), "w_sp")
                _bn(spec, r + ".1", cout)
            else:
                spec[r + ".conv1.weight"] = ((cout, 3, 3, 3, cout), "w_sp")
                _bn(spec, r + ".bn1", cout)
                spec[r + ".conv2.weight"] = ((cout, 3, 3, 3, cout), "w_sp")
                _bn(spec, r + ".bn2", cout)
            cin = cout
    spec[q + ".conv_out.0.weight"] = ((me["output_channels"], 3, 1, 1, cin), "w_sp")
    _bn(spec, q + ".conv_out.1", me["output_channels"])
    bb = cfg["lidar_encoder"]["pts_backbone"]
    cin = bb["in_channels"]
    for b, (cout, n) in enumerate(zip(bb["out_channels"], bb["layer_nums"])):
        for l in range(n + 1):
            _conv(spec, f"{p}.pts_backbone.blocks.{b}.{3 * l}", cout, cin, 3, bias=False)
            _bn(spec, f"{p}.pts_backbone.blocks.{b}.{3 * l + 1}", cout)
            cin = cout
    nk = cfg["lidar_encoder"]["pts_neck"]
    _conv(spec, p + ".pts_neck.deblocks.0.0", nk["out_channels"][0], nk["in_channels"][0], 1, bias=False)
    _bn(spec, p + ".pts_neck.deblocks.0.1", nk["out_channels"][0])
    spec[p + ".pts_neck.deblocks.1.0.weight"] = ((nk["in_channels"][1], nk["out_channels"][1], 2, 2), "w_deconv")
    _bn(spec, p + ".pts_neck.deblocks.1.1", nk["out_channels"][1])


def _fusion(spec):
    _lin(spec, "measurements_encoder.0", 128, 9)
    _lin(spec, "measurements_encoder.2", 128, 128)
    for name, cin in (("conv_cam", 256), ("conv_lidar", 512), ("conv_fusion", 512)):
        _conv(spec, f"{name}.0", 256, cin, 3, bias=False)
        _bn(spec, f"{name}.1", 256)
        _conv(spec, f"{name}.3", 256, 256, 3, bias=False)
        _bn(spec, f"{name}.4", 256)
    _conv(spec, "_256_to_32", 32, 256, 3)
    for name, c in (("MLP21", 32), ("MLP10", 64), ("MLP4", 128), ("MLP2", 256)):
        _conv(spec, f"{name}.conv1", 2 * c, c, 3)
        _bn(spec, f"{name}.bn1", 2 * c)
        _conv(spec, f"{name}.conv2", c, 2 * c, 3)
        _bn(spec, f"{name}.bn2", c)
        _conv(spec, f"{name}.se.fc1", c, c, 1)
        _conv(spec, f"{name}.se.fc2", c, c, 1)
    _conv(spec, "conv21_10", 64, 32, 3)
    _conv(spec, "conv10_4", 128, 64, 3)
    _conv(spec, "conv4_2", 256, 128, 3)
    _lin(spec, "output_fc.0", 512, 1024)
    _bn(spec, "output_fc.2", 512)
    _lin(spec, "output_fc.3", 256, 512)


def _mlp3(spec, p, dims, idx):
    for i, j in enumerate(idx):
        _lin(spec, f"{p}.{j}", dims[i + 1], dims[i])


def _decoder(spec, p, refine_num):
    spec[p + ".temporal_embedding"] = ((4, 128), "embed")
    spec[p + ".cams_embeds"] = ((4, 256), "embed")
    spec[p + ".static_embedding"] = ((4, 128), "embed")
    spec[p + ".level_embeds"] = ((4, 256), "embed")
    _mlp3(spec, p + ".join_traj", (384, 512, 512, 256), (0, 2, 4))
    _mlp3(spec, p + ".output_traj", (256, 512, 8), (0, 2))
    _mlp3(spec, p + ".join_ctrl", (384, 512, 512, 256), (0, 2, 4))
    for n in ("speed_branch", "value_branch_traj", "value_branch_ctrl"):
        _mlp3(spec, f"{p}.{n}", (256, 256, 256, 1), (0, 2, 4))
    _mlp3(spec, p + ".policy_head", (256, 512, 512), (0, 2))
    _mlp3(spec, p + ".dist_mu", (512, 512, 8), (0, 2))
    _mlp3(spec, p + ".dist_sigma", (512, 512, 8), (0, 2))
    for i in range(4):
        _conv(spec, f"{p}.fpn_linear{i}", 256, 256, 1)
    for L in range(refine_num):
        q = f"{p}.decoder_layers.{L}"
        g = q + ".prediction_module.spatial_gru"
        for n in ("conv_update", "conv_reset", "conv_state_tilde"):
            _conv(spec, f"{g}.{n}.0", 32, 38, 3)
            _conv(spec, f"{g}.{n}.2", 32, 32, 3)
        _conv(spec, g + ".conv_decoder.0", 32, 32, 3)
        _conv(spec, g + ".conv_decoder.2", 32, 32, 3)
        f = q + ".prediction_module.ffn"          # dead in the forward output (DEC:44-46)
        _conv(spec, f + ".0", 64, 32, 1)
        _conv(spec, f + ".2", 32, 64, 3)
        _conv(spec, f + ".4", 32, 32, 1)
        c = q + ".look_module.cam_look_module"
        _lin(spec, c + ".deformable_attention.sampling_offsets", 512, 256, kind="msda_off_w")
        spec[c + ".deformable_attention.sampling_offsets.bias"] = ((512,), "msda_off_b")
        _lin(spec, c + ".deformable_attention.attention_weights", 256, 256, kind="w_small")
        _lin(spec, c + ".deformable_attention.value_proj", 256, 256)
        _ln(spec, c + ".query_linear.0", 1543)
        _lin(spec, c + ".query_linear.1", 512, 1543)
        _lin(spec, c + ".query_linear.3", 256, 512)
        _ln(spec, c + ".ffn.norm", 256)
        _lin(spec, c + ".ffn.w_1", 1024, 256)
        _lin(spec, c + ".ffn.w_2", 256, 1024)
        _ln(spec, c + ".output_proj.0", 1024)
        _lin(spec, c + ".output_proj.1", 512, 1024)
        _lin(spec, c + ".output_proj.3", 256, 512)
        m = q + ".look_module"                    # dead LiDAR-look branch (DEC:186) + unused MLP
        _lin(spec, m + ".lidar_look_module_atten.0", 256, 134)
        _lin(spec, m + ".lidar_look_module_atten.2", 512, 256)
        _lin(spec, m + ".lidar_look_module_MLP.0", 128, 512)
        _lin(spec, m + ".lidar_look_module_MLP.3", 256, 1152)
        _lin(spec, m + ".look_feature_MLP.0", 512, 2048)
        _lin(spec, m + ".look_feature_MLP.2", 128, 512)
        _ln(spec, q + ".mlp.0", 1024)
        _lin(spec, q + ".mlp.1", 512, 1024)
        _lin(spec, q + ".mlp.4", 512, 512)
        _mlp3(spec, q + ".traj_offset_module", (514, 256, 64, 2), (0, 2, 4))
        _mlp3(spec, q + ".ctrl_offset_module", (516, 256, 64, 4), (0, 2, 4))
        _conv(spec, q + ".BEV_feat_update_module.0", 128, 2080, 3)
        _conv(spec, q + ".BEV_feat_update_module.2", 32, 128, 3)
        _lin(spec, q + ".flattened_BEV_feat_update_module.0", 512, 2304)
        _lin(spec, q + ".flattened_BEV_feat_update_module.2", 256, 512)


def param_spec(cfg, parts=("fusion", "img_encoder", "lidar_encoder", "decoder")):
    spec = OrderedDict()
    if "fusion" in parts:
        _fusion(spec)
    if "img_encoder" in parts:
        _lss(spec, "img_encoder", cfg)
    if "lidar_encoder" in parts:
        _lidar(spec, "lidar_encoder", cfg)
    if "decoder" in parts:
        _decoder(spec, "decoder", cfg["cfg"]["refine_num"])
    return spec


def _gen(name, seed):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


_SMALL_GAMMA_SUFFIX = (".bn3.weight",)


def _is_residual_tail_bn(name):
    if name.endswith(".bn3.weight") and "img_backbone" in name:
        return True
    if name.endswith(".bn2.weight") and (".depth_conv." in name or "encoder_layers" in name):
        return True
    return False


# variance gain per weight (He = 2.0 keeps ReLU stacks at constant scale; linear stacks such as the
# FPN / decoder heads use smaller gains so activations stay O(1..10) through ~100 layers)
_GAINS = (
    ("img_encoder.img_neck", 0.5),
    ("img_encoder.seg_net", 1.0),
    ("img_encoder.depth_net.depth_conv.5", 0.25),
    ("img_encoder.depth_net.context_conv", 0.5),
    ("img_encoder.merge_seg_and_image", 0.5),
    ("img_encoder.bev_multiframe_merge", 0.3),
    ("img_encoder.neck_conv", 0.5),
    ("decoder.", 1.0),
    ("measurements_encoder", 0.3),
)


def _gain_for(name):
    for prefix, g in _GAINS:
        if name.startswith(prefix):
            return g
    return 2.0


def init_tensor(name, shape, kind, seed, cfg):
    g = _gen(name, seed)
    if kind in ("w", "w_small", "w_deconv", "w_sp", "dcn_off_w", "msda_off_w"):
        if kind == "w_sp":
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
        elif kind == "w_deconv":
            fan_in = shape[0]                      # each output pixel sees Cin x 1 tap
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
        std = math.sqrt(_gain_for(name) / fan_in)
        if kind == "w_small":
            std *= 0.5
        if kind == "dcn_off_w":
            std = 0.5 / math.sqrt(fan_in)
        if kind == "msda_off_w":
            std = 0.5 / math.sqrt(fan_in)
        return torch.randn(shape, generator=g) * std
    if kind == "b":
        return torch.randn(shape, generator=g) * 0.05
    if kind == "dcn_off_b":
        return torch.randn(shape, generator=g) * 0.5
    if kind == "msda_off_b":
        # the reference's grid init (MSDA:403-417): 8 directions x (point index + 1) ...
        heads, levels, points = 8, 4, 8
        th = torch.arange(heads, dtype=torch.float32) * (2.0 * math.pi / heads)
        grid = torch.stack([th.cos(), th.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(heads, 1, 1, 2).repeat(1, levels, points, 1)
        for i in range(points):
            grid[:, :, i, :] *= i + 1
        # ... plus a small seeded jitter so the bias is not axis-symmetric
        return grid.reshape(-1) + torch.randn(shape, generator=g) * 0.1
    if kind == "bn_w":
        lo, hi = (0.1, 0.3) if _is_residual_tail_bn(name) else (0.5, 1.5)
        return lo + (hi - lo) * torch.rand(shape, generator=g)
    if kind == "bn_b":
        return torch.randn(shape, generator=g) * 0.1
    if kind == "bn_mean":
        if name.endswith("depth_net.bn.running_mean"):   # raw camera parameters are O(100)
            return torch.rand(shape, generator=g) * 300.0
        return torch.randn(shape, generator=g) * 0.1
    if kind == "bn_var":
        if name.endswith("depth_net.bn.running_var"):
            return (1.0 + 3.0 * torch.rand(shape, generator=g)) * 1e4
        return 0.5 + torch.rand(shape, generator=g)
    if kind == "bn_nbt":
        return torch.zeros(shape, dtype=torch.long)
    if kind == "ln_w":
        return 0.8 + 0.4 * torch.rand(shape, generator=g)
    if kind == "ln_b":
        return torch.randn(shape, generator=g) * 0.05
    if kind == "embed":
        return torch.randn(shape, generator=g) * 0.02
    enc = cfg["img_encoder"]
    rows = [enc["x_bound"], enc["y_bound"], enc["z_bound"]]
    if kind == "buf_voxel_size":
        return torch.Tensor([r[2] for r in rows])
    if kind == "buf_voxel_coord":
        return torch.Tensor([r[0] + r[2] / 2.0 for r in rows])
    if kind == "buf_voxel_num":
        return torch.LongTensor([(r[1] - r[0]) / r[2] for r in rows])
    if kind == "buf_frustum":
        from . import camera
        return camera.make_frustum(enc["final_dim"], enc["downsample_factor"], enc["d_bound"])
    raise KeyError(kind)


def init_params(cfg, seed=0, parts=("fusion", "img_encoder", "lidar_encoder", "decoder")):
    """OrderedDict name -> CPU tensor (the model's state_dict)."""
    sd = OrderedDict()
    for name, (shape, kind) in param_spec(cfg, parts).items():
        sd[name] = init_tensor(name, shape, kind, seed, cfg)
    return sd
