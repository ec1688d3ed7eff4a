"""Oracle (TEST INFRASTRUCTURE): PyTorch-CPU fp32 restatement of ThinkTwice's per-frame forward.

Functional style over a flat `state_dict` (same key names as the reference modules), channel-first
tensors like the reference.  Each function cites the reference lines it follows
(paths relative to /root/reference/open_loop_training/).

In-repo reference code (pinned by tests/golden/*.npz generated from the reference modules):
    code/encoder_decoder_framework.py   (EDF)   code/model_code/backbones/lss.py        (LSS)
    code/model_code/dense_heads/thinktwice_decoder.py (DEC)  .../multi_scale_deformable_attn_function.py (MSDA)
    code/model_code/dense_heads/utils.py (DHU)  code/utils.py (CU)   code/model_code/backbones/lidarnet.py (LID)
Third-party arithmetic NOT under /root/reference (restated from the pinned versions' documented
semantics, SURVEY.md Appendix C -- "parity unpinned"): mmdet ResNet/BasicBlock/PAFPN, mmcv DCN /
multi_scale_deformable_attn_pytorch / Voxelization, mmdet3d HardSimpleVFE / SparseEncoder /
SECOND / SECONDFPN, spconv.
"""

import numpy as np
import torch
import torch.nn.functional as F

from . import c_ref
from . import lss_geometry as geo


# ----------------------------------------------------------------------------- primitives
def conv(sd, p, x, stride=1, padding=0, dilation=1, groups=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, padding, dilation, groups)


def linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


# Training-mode switch (oracle/train_ref.py `train_mode()`): BatchNorm normalises with the statistics of the batch
# in front of it (running statistics are NOT updated here: the loss does not depend on them) and the one Dropout with
# p > 0 on the path (ASPP, LSS:91) draws its mask from torch's global RNG.  False = inference semantics.
TRAIN_MODE = False


def bn(sd, p, x, eps=1e-5):
    if TRAIN_MODE:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, eps)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, eps)


def layer_norm(sd, p, x):
    w = sd[p + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[p + ".bias"], 1e-5)


# ----------------------------------------------------------------------------- [3P] mmdet ResNet-50
def resnet50(sd, p, x):
    """mmdet ResNet(depth=50, out_indices 0-3), style='pytorch' (stride on the 3x3). LSS:401, CFG:141-148."""
    x = F.relu(bn(sd, p + ".bn1", conv(sd, p + ".conv1", x, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, blocks in enumerate((3, 4, 6, 3), start=1):
        for b in range(blocks):
            q = f"{p}.layer{li}.{b}"
            stride = 2 if (b == 0 and li > 1) else 1
            idt = x
            y = F.relu(bn(sd, q + ".bn1", conv(sd, q + ".conv1", x)))
            y = F.relu(bn(sd, q + ".bn2", conv(sd, q + ".conv2", y, stride, 1)))
            y = bn(sd, q + ".bn3", conv(sd, q + ".conv3", y))
            if b == 0:
                idt = bn(sd, q + ".downsample.1", conv(sd, q + ".downsample.0", x, stride))
            x = F.relu(y + idt)
        outs.append(x)
    return outs


# ----------------------------------------------------------------------------- [3P] mmdet PAFPN (== LSS:289-348)
def pafpn(sd, p, feats):
    n = len(feats)
    lat = [conv(sd, f"{p}.lateral_convs.{i}.conv", feats[i]) for i in range(n)]
    for i in range(n - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
    inter = [conv(sd, f"{p}.fpn_convs.{i}.conv", lat[i], 1, 1) for i in range(n)]
    for i in range(n - 1):
        inter[i + 1] = inter[i + 1] + conv(sd, f"{p}.downsample_convs.{i}.conv", inter[i], 2, 1)
    outs = [inter[0]]
    for i in range(1, n):
        outs.append(conv(sd, f"{p}.pafpn_convs.{i - 1}.conv", inter[i], 1, 1))
    return outs


# ----------------------------------------------------------------------------- [3P] mmdet BasicBlock
def basic_block(sd, p, x):
    y = F.relu(bn(sd, p + ".bn1", conv(sd, p + ".conv1", x, 1, 1)))
    y = bn(sd, p + ".bn2", conv(sd, p + ".conv2", y, 1, 1))
    return F.relu(y + x)


# ----------------------------------------------------------------------------- [3P] mmcv DCN (DeformConv2dPack v1)
def deform_im2col(x, offset, k=3, pad=1, dil=1):
    """Bilinear deformable columns: (B,C,H,W),(B,2*k*k,H,W) -> (B,C,k*k,H,W); zero outside,
    offset channel order [dy_0,dx_0,dy_1,dx_1,...] (deform_groups=1, stride 1)."""
    B, C, H, W = x.shape
    ys = torch.arange(H, dtype=x.dtype).view(1, H, 1)
    xs = torch.arange(W, dtype=x.dtype).view(1, 1, W)
    cols = []
    for t in range(k * k):
        i, j = t // k, t % k
        py = ys + (i * dil - pad) + offset[:, 2 * t]
        px = xs + (j * dil - pad) + offset[:, 2 * t + 1]
        gx = 2.0 * px / max(W - 1, 1) - 1.0
        gy = 2.0 * py / max(H - 1, 1) - 1.0
        grid = torch.stack([gx, gy], -1)
        cols.append(F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True))
    return torch.stack(cols, 2)


def dcn(sd, p, x, groups=4):
    off = conv(sd, p + ".conv_offset", x, 1, 1)
    cols = deform_im2col(x, off)                                   # (B,C,9,H,W)
    B, C, K, H, W = cols.shape
    w = sd[p + ".weight"]                                           # (Cout, C/groups, 3, 3)
    Cout = w.shape[0]
    cg, og = C // groups, Cout // groups
    out = []
    for g in range(groups):
        cg_cols = cols[:, g * cg:(g + 1) * cg].reshape(B, cg * K, H * W)
        wg = w[g * og:(g + 1) * og].reshape(og, cg * K)
        out.append(torch.matmul(wg, cg_cols))
    return torch.cat(out, 1).view(B, Cout, H, W)


# ----------------------------------------------------------------------------- LSS-owned blocks
def aspp(sd, p, x):
    """ASPP, LSS:49-118 (dilations 1/6/12/18; Dropout(0.5) LSS:91,110: identity in eval)."""
    def branch(name, padding, dilation):
        return F.relu(bn(sd, f"{p}.{name}.bn", conv(sd, f"{p}.{name}.atrous_conv", x, 1, padding, dilation)))
    x1 = branch("aspp1", 0, 1)
    x2 = branch("aspp2", 6, 6)
    x3 = branch("aspp3", 12, 12)
    x4 = branch("aspp4", 18, 18)
    x5 = F.adaptive_avg_pool2d(x, (1, 1))
    x5 = F.relu(bn(sd, p + ".global_avg_pool.2", conv(sd, p + ".global_avg_pool.1", x5)))
    x5 = F.interpolate(x5, size=x4.shape[2:], mode="bilinear", align_corners=True)
    y = torch.cat((x1, x2, x3, x4, x5), 1)
    y = F.relu(bn(sd, p + ".bn1", conv(sd, p + ".conv1", y)))
    return F.dropout(y, 0.5, True) if TRAIN_MODE else y


def depth_mlp_input(intrin, ida, sensor2ego):
    """(B*N, 22) camera-parameter vector of the KEY frame, DepthNet.forward LSS:206-231."""
    k = intrin[:, -1]
    a = ida[:, -1]
    s = sensor2ego[:, -1][..., :3, :]
    B, N = k.shape[:2]
    v = torch.stack([k[..., 0, 0], k[..., 1, 1], k[..., 0, 2], k[..., 1, 2],
                     a[..., 0, 0], a[..., 0, 1], a[..., 0, 3], a[..., 1, 0], a[..., 1, 1], a[..., 1, 3]], -1)
    return torch.cat([v, s.reshape(B, N, 12)], -1).reshape(B * N, 22)


def depth_net(sd, p, x, mlp_in):
    """DepthNet.forward LSS:205-240 -> cat[depth logits (D), context (256)]."""
    m = bn(sd, p + ".bn", mlp_in)
    x = F.relu(bn(sd, p + ".reduce_conv.1", conv(sd, p + ".reduce_conv.0", x, 1, 1)))

    def se(name, feat):
        v = linear(sd, f"{p}.{name}_mlp.fc2", F.relu(linear(sd, f"{p}.{name}_mlp.fc1", m)))[..., None, None]
        v = conv(sd, f"{p}.{name}_se.conv_expand", F.relu(conv(sd, f"{p}.{name}_se.conv_reduce", v)))
        return feat * torch.sigmoid(v)
    context = conv(sd, p + ".context_conv", se("context", x))
    d = se("depth", x)
    for i in range(3):
        d = basic_block(sd, f"{p}.depth_conv.{i}", d)
    d = aspp(sd, p + ".depth_conv.3", d)
    d = dcn(sd, p + ".depth_conv.4", d)
    d = conv(sd, p + ".depth_conv.5", d)
    return torch.cat([d, context], 1)


def unet(sd, p, feats):
    """UNet.forward LSS:275-282 on the 4 FPN maps."""
    e1, e2, e3, e4 = feats

    def up_layer(name, lo, skip):
        y = F.conv_transpose2d(lo, sd[f"{p}.{name}.up.weight"], sd[f"{p}.{name}.up.bias"], stride=2)
        y = torch.cat((y, skip), 1)
        return F.relu(conv(sd, f"{p}.{name}.conv_relu.0", y, 1, 1))
    d4 = up_layer("unet_layer4", e4, e3)
    d3 = up_layer("unet_layer3", d4, e2)
    d2 = up_layer("unet_layer2", d3, e1)
    d0 = F.interpolate(d2, scale_factor=2, mode="bilinear", align_corners=True)
    d0 = F.relu(conv(sd, p + ".unet_layer0.1", d0, 1, 1))
    d0 = conv(sd, p + ".unet_layer0.3", d0, 1, 1)
    return conv(sd, p + ".conv_last", d0)


def seg_to_feature(sd, p, seg):
    """seg_res_to_image_feature LSS:409-438."""
    x = seg
    for idx, (stride, pad) in zip((0, 3, 6, 9, 12, 15, 18),
                                  ((1, 0), (1, 0), (2, 1), (1, 0), (2, 1), (1, 0), (2, 1))):
        x = F.relu(bn(sd, f"{p}.{idx + 1}", conv(sd, f"{p}.{idx}", x, stride, pad)))
    return x


def rot_flip(x):
    """torch.rot90(torch.flip(x, [2]), 1, [2, 3])  (EDF:241,246)."""
    return torch.rot90(torch.flip(x, dims=[2]), 1, dims=[2, 3])


def lift_splat(depth_logits, context, geom_idx, voxel_num, num_cams):
    """softmax (x) context -> permute -> voxel pooling (LSS:583,593-615,629-632; VP.cu:9-36)."""
    BN, D, H, W = depth_logits.shape
    C = context.shape[1]
    B = BN // num_cams
    vol = depth_logits.softmax(1).unsqueeze(1) * context.unsqueeze(2)          # (BN,C,D,H,W)
    vol = vol.reshape(B, num_cams, C, D, H, W).permute(0, 1, 3, 4, 5, 2).contiguous()
    if torch.is_grad_enabled() and vol.requires_grad:
        # differentiable form for gradient checks: index_add over the in-range points == the reference's autograd
        # Function (forward VP.cu:9-36, backward = gather of grad_out by pos_memo, voxel_pooling.py:57-69)
        vx, vy, vz = (int(v) for v in voxel_num)
        g = geom_idx.reshape(B, -1, 3).long()
        ok = (g[..., 0] >= 0) & (g[..., 0] < vx) & (g[..., 1] >= 0) & (g[..., 1] < vy) & (g[..., 2] >= 0) & (g[..., 2] < vz)
        lin = (torch.arange(B).view(B, 1) * vy + g[..., 1]) * vx + g[..., 0]
        flat = torch.zeros(B * vy * vx, C).index_add(0, lin[ok], vol.reshape(B, -1, C)[ok])
        return flat.view(B, vy, vx, C).permute(0, 3, 1, 2).contiguous()
    out, _ = c_ref.voxel_pool_fwd(geom_idx.reshape(B, -1, 3).numpy(), vol.reshape(B, -1, C).numpy(),
                                  [int(v) for v in voxel_num], acc64=False, want_pos_memo=False)
    return torch.from_numpy(out).permute(0, 3, 1, 2).contiguous()               # (B,C,Y,X)


def lss_single_sweep(sd, p, cfg, imgs, mats, geom_idx):
    """LSS._forward_single_sweep LSS:542-621 for one sweep: imgs (B,N,3,H,W)."""
    B, N = imgs.shape[:2]
    x = imgs.reshape(B * N, *imgs.shape[2:])
    fpn = pafpn(sd, p + ".img_neck", resnet50(sd, p + ".img_backbone", x))
    src = conv(sd, p + ".neck_conv", fpn[2])
    D = sd[p + ".frustum"].shape[0]
    df = depth_net(sd, p + ".depth_net", src, depth_mlp_input(*mats))
    depth, ctx = df[:, :D], df[:, D:D + 256]
    seg = unet(sd, p + ".seg_net", fpn)
    segf = seg_to_feature(sd, p + ".seg_res_to_image_feature", seg.detach())       # LSS:589 seg_output.detach()
    ctx = conv(sd, p + ".merge_seg_and_image", torch.cat((ctx, segf), 1), 1, 1)
    bev = lift_splat(depth, ctx, geom_idx, sd[p + ".voxel_num"], N)
    return {"bev": bev, "fpn_feats": fpn, "seg": seg, "depth": depth, "context": ctx}


def lss_forward(sd, p, cfg, img, img_metas):
    """LSS.forward LSS:635-724. img (B,T,N,3,H,W); key frame = last T index.  Both sweeps use the
    KEY frame's matrices (sweep loop passes index -1, LSS:712-714 / SURVEY A10 quirk)."""
    intr, ida, s2e, lidar2img, cur_ida = geo.assemble_camera_mats(img_metas)
    geom = geo.get_geometry(sd[p + ".frustum"], s2e[:, -1], intr[:, -1], ida[:, -1])
    idx = geo.voxel_index(geom, sd[p + ".voxel_coord"], sd[p + ".voxel_size"])
    mats = (intr, ida, s2e)
    key = lss_single_sweep(sd, p, cfg, img[:, -1], mats, idx)
    bevs = [key["bev"]]
    T = img.shape[1]
    for s in range(1, T):
        with torch.no_grad():                                                       # LSS:711 older sweeps carry no grad
            bevs.append(lss_single_sweep(sd, p, cfg, img[:, T - 1 - s], mats, idx)["bev"])
    bev = torch.cat(bevs, 1)
    if T > 1:
        bev = F.conv2d(bev, sd[p + ".bev_multiframe_merge.weight"], None, 1, 1)
    return {"bev": bev, "seg": key["seg"], "fpn_feats": key["fpn_feats"], "lidar2img": lidar2img,
            "ida_mat": cur_ida, "depth": key["depth"], "context": key["context"], "geom_idx": idx}


# ----------------------------------------------------------------------------- LiDAR encoder ([3P])
def hard_voxelize(points, voxel_size, pc_range, max_points=10, max_voxels=160000):
    """mmcv Voxelization (hard, deterministic) for ONE sample: points (Np,5) ->
    voxels (M,max_points,5) zero padded, coors (M,3) as (z,y,x), num_points (M,)."""
    pts = points.numpy().astype(np.float32)
    vs = np.asarray(voxel_size, np.float32)
    lo = np.asarray(pc_range[:3], np.float32)
    hi = np.asarray(pc_range[3:], np.float32)
    grid = np.round((hi - lo) / vs).astype(np.int64)
    c = np.floor((pts[:, :3] - lo) / vs).astype(np.int64)
    ok = ((c >= 0) & (c < grid)).all(1)
    idx_pts = np.nonzero(ok)[0]
    c = c[ok]
    key = (c[:, 2] * grid[1] + c[:, 1]) * grid[0] + c[:, 0]
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")            # voxels in order of first appearance
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    vox_of_pt = rank[inv]
    keep_vox = vox_of_pt < max_voxels
    M = min(len(uniq), max_voxels)
    voxels = np.zeros((M, max_points, pts.shape[1]), np.float32)
    num = np.zeros((M,), np.int64)
    # points are visited in order; each voxel keeps its first `max_points`
    srt = np.argsort(vox_of_pt, kind="stable")
    v_sorted = vox_of_pt[srt]
    starts = np.searchsorted(v_sorted, np.arange(M))
    pos_in_vox = np.arange(len(srt)) - starts[np.minimum(v_sorted, M - 1)]
    sel = (pos_in_vox < max_points) & keep_vox[srt]
    voxels[v_sorted[sel], pos_in_vox[sel]] = pts[idx_pts[srt[sel]]]
    np.add.at(num, v_sorted[sel], 1)
    coors = np.zeros((M, 3), np.int64)
    first_pt = first[order][:M]
    coors[:, 0], coors[:, 1], coors[:, 2] = c[first_pt, 2], c[first_pt, 1], c[first_pt, 0]
    return torch.from_numpy(voxels), torch.from_numpy(coors), torch.from_numpy(num)


class SparseT:
    def __init__(self, feats, coords, shape, batch):
        self.feats, self.coords, self.shape, self.batch = feats, coords, list(shape), batch

    def index_volume(self, pad):
        D, H, W = self.shape
        vol = torch.full((self.batch, D + 2 * pad[0], H + 2 * pad[1], W + 2 * pad[2]), -1, dtype=torch.int64)
        c = self.coords
        vol[c[:, 0], c[:, 1] + pad[0], c[:, 2] + pad[1], c[:, 3] + pad[2]] = torch.arange(c.shape[0])
        return vol


def subm_conv3d(st, w):
    """spconv SubMConv3d k3 p1 (outputs only at active input sites); w (Cout,3,3,3,Cin)."""
    vol = st.index_volume((1, 1, 1))
    c = st.coords
    out = torch.zeros(c.shape[0], w.shape[0])
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                nb = vol[c[:, 0], c[:, 1] + kz, c[:, 2] + ky, c[:, 3] + kx]
                m = nb >= 0
                if m.any():
                    out[m] += st.feats[nb[m]] @ w[:, kz, ky, kx, :].t()
    return SparseT(out, c, st.shape, st.batch)


def sparse_conv3d(st, w, stride, pad):
    """spconv SparseConv3d: output site active iff any input lies in its receptive field."""
    ks = w.shape[1:4]
    oshape = [(st.shape[d] + 2 * pad[d] - ks[d]) // stride[d] + 1 for d in range(3)]
    c = st.coords
    cand_rows, cand_out, cand_tap = [], [], []
    for kz in range(ks[0]):
        for ky in range(ks[1]):
            for kx in range(ks[2]):
                num = torch.stack([c[:, 1] + pad[0] - kz, c[:, 2] + pad[1] - ky, c[:, 3] + pad[2] - kx], 1)
                s = torch.tensor(stride)
                ok = (num % s == 0).all(1)
                o = num // s
                ok &= ((o >= 0) & (o < torch.tensor(oshape))).all(1)
                idx = torch.nonzero(ok)[:, 0]
                cand_rows.append(idx)
                cand_out.append(torch.cat([c[idx, :1], o[idx]], 1))
                cand_tap.append(torch.full((idx.shape[0],), (kz * ks[1] + ky) * ks[2] + kx))
    rows = torch.cat(cand_rows)
    outs = torch.cat(cand_out)
    taps = torch.cat(cand_tap)
    key = ((outs[:, 0] * oshape[0] + outs[:, 1]) * oshape[1] + outs[:, 2]) * oshape[2] + outs[:, 3]
    uniq, inv = torch.unique(key, return_inverse=True)
    oc = torch.stack([uniq // (oshape[0] * oshape[1] * oshape[2]), (uniq // (oshape[1] * oshape[2])) % oshape[0],
                      (uniq // oshape[2]) % oshape[1], uniq % oshape[2]], 1)
    out = torch.zeros(uniq.shape[0], w.shape[0])
    wf = w.reshape(w.shape[0], -1, w.shape[-1])
    for t in range(wf.shape[1]):
        m = taps == t
        if m.any():
            out.index_add_(0, inv[m], st.feats[rows[m]] @ wf[:, t, :].t())
    return SparseT(out, oc, oshape, st.batch)


def _sp_bn_relu(sd, p, st, relu=True):
    if TRAIN_MODE:
        f = F.batch_norm(st.feats, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, 1e-3)
    else:
        f = F.batch_norm(st.feats, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                         sd[p + ".bias"], False, 0.0, 1e-3)
    return SparseT(F.relu(f) if relu else f, st.coords, st.shape, st.batch)


def sparse_encoder(sd, p, cfg, feats, coors, batch):
    """mmdet3d SparseEncoder (LID:28-58 wraps it): conv_input, 4 stages, conv_out, dense, view."""
    me = cfg["lidar_encoder"]["pts_middle_encoder"]
    st = SparseT(feats, coors.long(), me["sparse_shape"], batch)
    st = _sp_bn_relu(sd, p + ".conv_input.1", subm_conv3d(st, sd[p + ".conv_input.0.weight"]))
    chans, pads = me["encoder_channels"], me["encoder_paddings"]
    for i, blocks in enumerate(chans):
        for j in range(len(blocks)):
            r = f"{p}.encoder_layers.encoder_layer{i + 1}.{j}"
            if j == len(blocks) - 1 and i != len(chans) - 1:
                pd = pads[i][j]
                pd = [pd] * 3 if isinstance(pd, int) else list(pd)
                st = _sp_bn_relu(sd, r + ".1", sparse_conv3d(st, sd[r + ".0.weight"], (2, 2, 2), pd))
            else:   # SparseBasicBlock
                y = _sp_bn_relu(sd, r + ".bn1", subm_conv3d(st, sd[r + ".conv1.weight"]))
                y = _sp_bn_relu(sd, r + ".bn2", subm_conv3d(y, sd[r + ".conv2.weight"]), relu=False)
                st = SparseT(F.relu(y.feats + st.feats), st.coords, st.shape, st.batch)
    st = _sp_bn_relu(sd, p + ".conv_out.1", sparse_conv3d(st, sd[p + ".conv_out.0.weight"], (2, 1, 1), (0, 0, 0)))
    D, H, W = st.shape
    dense = torch.zeros(batch, st.feats.shape[1], D, H, W)
    c = st.coords
    dense[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = st.feats
    return dense.view(batch, -1, H, W)


def lidar_net(sd, p, cfg, points):
    """LidarNet.forward LID:87-96: points (B,Np,5) -> [(B,512,84,84)]."""
    vl = cfg["lidar_encoder"]["pts_voxel_layer"]
    me = cfg["lidar_encoder"]["pts_middle_encoder"]
    feats, coors = [], []
    for b in range(points.shape[0]):
        v, c, n = hard_voxelize(points[b], vl["voxel_size"], vl["point_cloud_range"], vl["max_num_points"],
                                vl["max_voxels"][1])
        keep = c[:, 0] < me["sparse_shape"][0]        # z bins beyond the declared sparse grid (quirk 9)
        v, c, n = v[keep], c[keep], n[keep]
        feats.append(v.sum(1) / n[:, None].float())    # HardSimpleVFE
        coors.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int64), c], 1))
    feats, coors = torch.cat(feats), torch.cat(coors)
    x = sparse_encoder(sd, p + ".pts_middle_encoder", cfg, feats, coors, points.shape[0])
    bb = cfg["lidar_encoder"]["pts_backbone"]
    outs = []
    for b, (n, s) in enumerate(zip(bb["layer_nums"], bb["layer_strides"])):
        for l in range(n + 1):
            q = f"{p}.pts_backbone.blocks.{b}"
            x = F.relu(bn(sd, f"{q}.{3 * l + 1}", conv(sd, f"{q}.{3 * l}", x, s if l == 0 else 1, 1), 1e-3))
        outs.append(x)
    u0 = F.relu(bn(sd, p + ".pts_neck.deblocks.0.1", conv(sd, p + ".pts_neck.deblocks.0.0", outs[0]), 1e-3))
    u1 = F.conv_transpose2d(outs[1], sd[p + ".pts_neck.deblocks.1.0.weight"], None, stride=2)
    u1 = F.relu(bn(sd, p + ".pts_neck.deblocks.1.1", u1, 1e-3))
    return [torch.cat([u0, u1], 1)]


# ----------------------------------------------------------------------------- BEV fusion + flatten (EDF)
def se_basic_block(sd, p, x):
    """SEBasicBlock CU:99-121 with SEModule CU:84-96 (pool = mean/2 + max/2)."""
    y = F.relu(bn(sd, p + ".bn1", conv(sd, p + ".conv1", x, 1, 1)))
    y = F.relu(bn(sd, p + ".bn2", conv(sd, p + ".conv2", y, 1, 1)))
    s = 0.5 * y.mean((2, 3), keepdim=True) + 0.5 * y.amax((2, 3), keepdim=True)
    s = conv(sd, p + ".se.fc2", F.relu(conv(sd, p + ".se.fc1", s)))
    return F.relu(y * torch.sigmoid(s) + x)


def flatten_tail(sd, f21):
    """conv21_10 ... output_fc (EDF:228-234 == DEC:405-415 grid2feat)."""
    f10 = se_basic_block(sd, "MLP10", F.relu(conv(sd, "conv21_10", f21, 2)))
    f4 = se_basic_block(sd, "MLP4", F.relu(conv(sd, "conv10_4", f10, 2)))
    f2 = se_basic_block(sd, "MLP2", F.relu(conv(sd, "conv4_2", f4, 1)))
    h = F.relu(linear(sd, "output_fc.0", f2.flatten(1)))
    h = bn(sd, "output_fc.2", h)
    return F.relu(linear(sd, "output_fc.3", h)), [f10, f4, f2]


def fusion(sd, cam_bev, lidar_feat):
    """EncoderDecoder.get_fusion_feat EDF:213-235."""
    def two_conv(name, x, stride, last_relu):
        y = F.relu(bn(sd, name + ".1", conv(sd, name + ".0", x, stride, 1)))
        y = bn(sd, name + ".4", conv(sd, name + ".3", y, stride, 1))
        return F.relu(y) if last_relu else y
    cam = F.relu(two_conv("conv_cam", cam_bev, 1, False) + cam_bev)
    pts = two_conv("conv_lidar", lidar_feat, 2, True)
    bev = F.relu(two_conv("conv_fusion", torch.cat([cam, pts], 1), 1, False) + cam + pts)
    f21 = se_basic_block(sd, "MLP21", F.relu(conv(sd, "_256_to_32", bev, 1, 1)))
    flat, mids = flatten_tail(sd, f21)
    return flat, f21, [None, None, f21] + mids


# ----------------------------------------------------------------------------- decoder (DEC / MSDA / DHU)
def mlp_seq(sd, p, x, idx, last_act=False):
    for n, j in enumerate(idx):
        x = linear(sd, f"{p}.{j}", x)
        if n < len(idx) - 1 or last_act:
            x = F.relu(x)
    return x


def spatial_gru(sd, p, inp6, state, steps=4):
    """SpatialGRU.forward/gru_cell DHU:82-106 with a time-constant input per step index."""
    def two(name, x):
        return conv(sd, f"{p}.{name}.2", F.relu(conv(sd, f"{p}.{name}.0", x, 1, 1)), 1, 1)
    outs = []
    for t in range(steps):
        x = inp6[:, t]
        xs = torch.cat([x, state], 1)
        u = torch.sigmoid(two("conv_update", xs))
        r = torch.sigmoid(two("conv_reset", xs))
        cand = two("conv_state_tilde", torch.cat([x, (1.0 - r) * state], 1))
        state = (1.0 - u) * state + u * cand
        outs.append(two("conv_decoder", state))
    return torch.stack(outs, 1)


def msda_core(value, spatial_shapes, loc, attw):
    """[3P] mmcv multi_scale_deformable_attn_pytorch: value (bs,S,heads,dh); loc (bs,Q,heads,L,P,2);
    attw (bs,Q,heads,L,P) -> (bs,Q,heads*dh)."""
    bs, _, heads, dh = value.shape
    _, Q, _, L, P, _ = loc.shape
    grids = 2 * loc - 1
    start = 0
    sampled = []
    for lvl, (h, w) in enumerate(spatial_shapes):
        v = value[:, start:start + h * w].flatten(2).transpose(1, 2).reshape(bs * heads, dh, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)                    # (bs*heads,Q,P,2)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
        start += h * w
    a = attw.transpose(1, 2).reshape(bs * heads, 1, Q, L * P)
    out = (torch.stack(sampled, -2).flatten(-2) * a).sum(-1).view(bs, heads * dh, Q)
    return out.transpose(1, 2).contiguous()


def project_queries(pts3d, lidar2img, ida_mat, img_hw):
    """LookModule.obtain_cam_ref_points_query DEC:89-113: (B,Q,3) -> ref (B,N,Q,2) in [0,1], mask (B,N,Q)."""
    B, Q, _ = pts3d.shape
    N = lidar2img.shape[1]
    hom = torch.cat([pts3d, torch.ones_like(pts3d[..., :1])], -1)
    cam = torch.matmul(lidar2img.view(B, N, 1, 4, 4), hom.view(B, 1, Q, 4, 1)).squeeze(-1)
    eps = 1e-5
    z = torch.maximum(cam[..., 2:3], torch.ones_like(cam[..., 2:3]) * eps)
    cam2 = cam.clone()
    cam2[..., 0:2] = cam[..., 0:2] / z
    img = torch.matmul(ida_mat.view(B, N, 1, 4, 4), cam2.unsqueeze(-1)).squeeze(-1)
    mask = img[..., 2] > eps
    ref = img[..., :2].clone()
    ref[..., 0] = ref[..., 0] / img_hw[1]
    ref[..., 1] = ref[..., 1] / img_hw[0]
    mask = mask & (ref[..., 1] > 0.0) & (ref[..., 1] < 1.0) & (ref[..., 0] < 1.0) & (ref[..., 0] > 0.0)
    return ref, mask


def look_module(sd, p, cfg, wp, ctrl_sp, meas, flat, lidar2img, ida_mat, mlvl, value_in, shapes,
                temporal, static):
    """LookModule.forward DEC:154-187 (camera branch; LiDAR branch is zeroed at DEC:186) +
    SpatialCrossAttention.forward MSDA:279-344 with its batch-coupled normalisation bug."""
    B = wp.shape[0]
    N = 4
    static_pt = torch.tensor([[5.0, 0.0], [0.0, -5.0], [0.0, 5.0], [-5.0, 0.0]]).unsqueeze(0).expand(B, 4, 2)
    look = torch.cat([wp, static_pt], 1)                                          # (B,8,2)
    zs = torch.linspace(-4, 10, 15, dtype=torch.float64).to(look.dtype)
    p3 = torch.cat([look.unsqueeze(2).expand(B, 8, 15, 2), zs.view(1, 1, 15, 1).expand(B, 8, 15, 1)], -1)
    p3 = p3.reshape(B, 120, 3)
    ctrl_q = torch.cat([ctrl_sp.unsqueeze(2).expand(B, 4, 15, 4).reshape(B, 60, 4), torch.zeros(B, 60, 4)], 1)
    emb = torch.cat([temporal.unsqueeze(1).expand(4, 15, 128).reshape(60, 128),
                     static.unsqueeze(1).expand(4, 15, 128).reshape(60, 128)], 0).unsqueeze(0).expand(B, 120, 128)
    query = torch.cat([ctrl_q, p3, emb, meas.unsqueeze(1).expand(B, 120, 128),
                       flat.unsqueeze(1).expand(B, 120, 256)], -1)                 # (B,120,519)
    ref, mask = project_queries(p3, lidar2img, ida_mat, cfg["cfg"]["img_size"])    # (B,N,120,2/·)
    grid = ref.reshape(B * N, 120, 1, 2) * 2 - 1.0
    samp = [F.grid_sample(f, grid, mode="bilinear", padding_mode="zeros", align_corners=False)[..., 0]
            for f in mlvl]                                                          # (BN,256,120) x4
    samp = torch.stack(samp, -1).permute(0, 2, 1, 3).reshape(B, N, 120, 1024)      # channel-major, level-minor
    # left-pack the valid queries of every (sample, camera); pad with zero query / zero ref
    order = torch.argsort((~mask).to(torch.int8), dim=-1, stable=True)             # (B,N,120)
    count = mask.sum(-1)
    max_len = int(count.max())
    slot_ok = torch.arange(120).view(1, 1, 120) < count.unsqueeze(-1)
    qfull = torch.cat([query.unsqueeze(1).expand(B, N, 120, 519), samp], -1)
    qpack = torch.gather(qfull, 2, order.unsqueeze(-1).expand(B, N, 120, 1543)) * slot_ok.unsqueeze(-1)
    rpack = torch.gather(ref, 2, order.unsqueeze(-1).expand(B, N, 120, 2)) * slot_ok.unsqueeze(-1)
    qpack, rpack = qpack[:, :, :max_len], rpack[:, :, :max_len]
    c = p + ".cam_look_module"
    q = F.gelu(linear(sd, c + ".query_linear.3", F.gelu(linear(sd, c + ".query_linear.1",
                                                               layer_norm(sd, c + ".query_linear.0", qpack)))))
    d = c + ".deformable_attention"
    value = linear(sd, d + ".value_proj", value_in).view(B * N, -1, 8, 32)
    q2 = q.reshape(B * N, max_len, 256)
    off = linear(sd, d + ".sampling_offsets", q2).view(B * N, max_len, 8, 4, 8, 2)
    aw = linear(sd, d + ".attention_weights", q2).view(B * N, max_len, 8, 32).softmax(-1).view(B * N, max_len, 8, 4, 8)
    norm = torch.tensor([[w, h] for (h, w) in shapes], dtype=q.dtype)
    loc = rpack.reshape(B * N, max_len, 1, 1, 1, 2) + off / norm.view(1, 1, 1, 4, 1, 2)
    att = msda_core(value, shapes, loc, aw).view(B, N, max_len, 256)
    f = c + ".ffn"
    att = linear(sd, f + ".w_2", F.gelu(linear(sd, f + ".w_1", layer_norm(sd, f + ".norm", att)))) + att
    # MSDA:338-341: `indexes` is list[cam][sample] => len()==B: first B slots zeroed, all divided by B
    att = att.clone()
    att[:, :, :B] = 0
    att = att / max(B, 1.0)
    att = att.sum(-2).reshape(B, N * 256)
    o = c + ".output_proj"
    out = linear(sd, o + ".3", F.gelu(linear(sd, o + ".1", layer_norm(sd, o + ".0", att))))
    return out, {"max_len": max_len, "count": count}


def decoder_layer(sd, p, cfg, bev, wp, ctrl, meas, flat, look_args):
    """ThinkTwiceDecoderLayer.forward DEC:236-260."""
    B = bev.shape[0]
    sp = F.softplus(ctrl)
    inp = torch.cat([wp, sp], 2)[..., None, None].expand(B, 4, 6, bev.shape[2], bev.shape[3])
    fut = spatial_gru(sd, p + ".prediction_module.spatial_gru", inp, bev)          # (B,4,32,21,21)
    fflat, _ = flatten_tail(sd, fut.reshape(B * 4, *bev.shape[1:]))
    fflat = fflat.view(B, 4, 256)
    temporal, static = sd["decoder.temporal_embedding"], sd["decoder.static_embedding"]
    lk, info = look_module(sd, p + ".look_module", cfg, wp, sp, meas, flat, *look_args, temporal, static)
    look = torch.cat([lk.unsqueeze(1).expand(B, 4, 256), torch.zeros(B, 4, 256)], -1)   # DEC:176,186
    h = torch.cat([fflat, look, temporal.unsqueeze(0).expand(B, 4, 128), meas.unsqueeze(1).expand(B, 4, 128)], -1)
    h = layer_norm(sd, p + ".mlp.0", h)
    h = F.relu(linear(sd, p + ".mlp.4", F.relu(linear(sd, p + ".mlp.1", h))))       # (B,4,512)
    d_wp = mlp_seq(sd, p + ".traj_offset_module", torch.cat([wp, h], -1), (0, 2, 4))
    d_ctrl = mlp_seq(sd, p + ".ctrl_offset_module", torch.cat([ctrl, h], -1), (0, 2, 4))
    hb = h.reshape(B, 2048)
    x = torch.cat([bev, hb[..., None, None].expand(B, 2048, bev.shape[2], bev.shape[3])], 1)
    nb = conv(sd, p + ".BEV_feat_update_module.2", F.relu(conv(sd, p + ".BEV_feat_update_module.0", x, 1, 1)), 1, 1) + bev
    nf = mlp_seq(sd, p + ".flattened_BEV_feat_update_module", torch.cat([flat, hb], -1), (0, 2)) + flat
    return d_wp, d_ctrl, fut, nb, nf, info


def decoder_forward(sd, cfg, flat, bev, meas, lidar2img, ida_mat, fpn_feats, teacher=None):
    """ThinkTwiceDecoder.forward DEC:419-489; with `teacher` (dict of expert waypoints / Beta parameters) also the
    teacher-forcing pass DEC:491-533: the same five layers run a second time from the encoder's BEV state with the
    EXPERT waypoints and inv_softplus(expert control) as constant inputs of every layer."""
    p = "decoder"
    out = {"bev_feature": bev}
    out["pred_speed"] = mlp_seq(sd, p + ".speed_branch", flat, (0, 2, 4))
    fm = torch.cat([flat, meas], 1)
    jt = mlp_seq(sd, p + ".join_traj", fm, (0, 2, 4), last_act=True)
    out["pred_value_traj"] = mlp_seq(sd, p + ".value_branch_traj", jt, (0, 2, 4))
    out["pred_features_traj"] = jt
    wps = [mlp_seq(sd, p + ".output_traj", jt, (0, 2)).view(-1, 4, 2)]
    jc = mlp_seq(sd, p + ".join_ctrl", fm, (0, 2, 4), last_act=True)
    out["pred_value_ctrl"] = mlp_seq(sd, p + ".value_branch_ctrl", jc, (0, 2, 4))
    out["pred_features_ctrl"] = jc
    pol = mlp_seq(sd, p + ".policy_head", jc, (0, 2), last_act=True)
    mu = mlp_seq(sd, p + ".dist_mu", pol, (0, 2)).view(-1, 4, 2)
    sg = mlp_seq(sd, p + ".dist_sigma", pol, (0, 2)).view(-1, 4, 2)
    ctrls = [torch.cat([mu, sg], -1)]
    # FPN -> values (DEC:446-450, transform_fpn_feats DEC:381-401)
    mlvl = [conv(sd, f"{p}.fpn_linear{i}", fpn_feats[i]) for i in range(4)]
    B = flat.shape[0]
    shapes = [tuple(f.shape[2:]) for f in mlvl]
    vals = []
    for lvl, f in enumerate(mlvl):
        v = f.view(B, 4, 256, -1).permute(0, 1, 3, 2)                               # (B,cam,HW,256)
        v = v + sd[p + ".cams_embeds"].view(1, 4, 1, 256) + sd[p + ".level_embeds"][lvl].view(1, 1, 1, 256)
        vals.append(v)
    value_in = torch.cat(vals, 2).reshape(B * 4, -1, 256)
    look_args = (lidar2img, ida_mat, mlvl, value_in, shapes)
    cur_bev, cur_flat = bev.clone(), flat.clone()
    s_bev, s_flat, s_fut, infos = [], [], [], []
    for L in range(cfg["cfg"]["refine_num"]):
        wp_in, ctrl_in = wps[-1].detach(), ctrls[-1].detach()                      # DEC:429-430
        d_wp, d_ctrl, fut, cur_bev, cur_flat, info = decoder_layer(
            sd, f"{p}.decoder_layers.{L}", cfg, cur_bev, wp_in, ctrl_in, meas, cur_flat, look_args)
        wps.append(d_wp + wp_in)
        ctrls.append(d_ctrl + ctrl_in)
        s_bev.append(cur_bev)
        s_flat.append(cur_flat)
        s_fut.append(fut)
        infos.append(info)
    out["refine_flattned_BEV_feature"] = torch.stack(s_flat, 1)
    out["refine_BEV_feature"] = torch.stack(s_bev, 1)
    # DEC:481: the (B,R,4,...) stack is re-VIEWED as (B,4,R,...) and transposed -- a memory
    # reinterpretation (slot [b,r,t] holds stack[b].flatten(0,1)[t*R + r]), reproduced as is.
    R = len(s_fut)
    out["refine_future_BEV_feature"] = torch.stack(s_fut, 1).view(B, 4, R, *s_fut[0].shape[2:]).transpose(1, 2)
    wp_all = torch.stack(wps, 1)
    ct = torch.clamp(F.softplus(torch.stack(ctrls, 1)), min=1e-3)
    out["pred_wp"] = wp_all
    out["mu_branches"], out["sigma_branches"] = ct[:, :, 0, :2], ct[:, :, 0, 2:]
    out["future_mu"], out["future_sigma"] = ct[:, :, 1:, :2], ct[:, :, 1:, 2:]
    out["_look_info"] = infos
    if teacher is not None:
        t_wp = teacher["waypoints"].float()
        cur = torch.cat([teacher["action_mu"], teacher["action_sigma"]], -1).unsqueeze(1)
        fut = torch.cat([torch.stack(teacher["future_action_mu"][:-1], 1),
                         torch.stack(teacher["future_action_sigma"][:-1], 1)], -1)
        sp = torch.cat([cur, fut], 1).float()
        t_ctrl = sp + torch.log(-torch.expm1(-sp))                       # inv_softplus, DEC:22-23
        cur_bev, cur_flat = bev.clone(), flat.clone()
        t_dwp, t_dctrl, t_bev, t_flat, t_fut = [], [], [], [], []
        for L in range(cfg["cfg"]["refine_num"]):
            d_wp, d_ctrl, futL, cur_bev, cur_flat, _ = decoder_layer(
                sd, f"{p}.decoder_layers.{L}", cfg, cur_bev, t_wp, t_ctrl, meas, cur_flat, look_args)
            t_dwp.append(d_wp)
            t_dctrl.append(d_ctrl)
            t_bev.append(cur_bev)
            t_flat.append(cur_flat)
            t_fut.append(futL)
        out["teacher_pred_wp_offset"] = torch.stack(t_dwp, 1)
        out["teacher_pred_ctrl_offset_lis"] = torch.stack(t_dctrl, 1)
        out["teacher_future_BEV_feature"] = torch.stack(t_fut, 1)
        out["teacher_refine_flattned_BEV_feature"] = torch.stack(t_flat, 1)
        out["teacher_refine_BEV_feature"] = torch.stack(t_bev, 1)
    return out


def measurement_feat(sd, batch):
    """EDF:198,242-243."""
    speed = batch["speed"].float().view(-1, 1) / 12.0
    state = torch.cat([speed, batch["target_point"].float(), batch["target_command"].float()], -1)
    return F.relu(linear(sd, "measurements_encoder.2", F.relu(linear(sd, "measurements_encoder.0", state))))


def forward_inference(sd, cfg, batch, return_intermediates=False):
    """EncoderDecoder.forward_inference EDF:194-210."""
    cam = lss_forward(sd, "img_encoder", cfg, batch["img"], batch["img_metas"])
    cam_bev = rot_flip(cam["bev"])
    meas = measurement_feat(sd, batch)
    lid = [rot_flip(t) for t in lidar_net(sd, "lidar_encoder", cfg, batch["points"][:, -1])]
    flat, bev32, mids = fusion(sd, cam_bev, lid[0])
    pred = decoder_forward(sd, cfg, flat, bev32, meas, cam["lidar2img"], cam["ida_mat"], cam["fpn_feats"])
    if return_intermediates:
        pred["_cam"] = cam
        pred["_cam_bev"] = cam_bev
        pred["_lidar_bev"] = lid[0]
        pred["_flat"] = flat
        pred["_meas"] = meas
    return pred
