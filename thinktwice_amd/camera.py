"""Host-side camera metadata plumbing of the LSS encoder (tiny 4x4 matrices, no image data).

Mirrors what LSS.__init__/LSS.forward do on the host before any kernel runs:
  buffers voxel_size / voxel_coord / voxel_num / frustum   backbones/lss.py:386-398,454-471
  img_metas -> intrin / ida / sensor2ego tensors            backbones/lss.py:667-687,704-707
  the matrices get_geometry applies                         backbones/lss.py:496,502
The per-point work (501,760 points per sample) runs in tt_frustum_voxel_index.
"""
import torch


class VoxelGrid:
    def __init__(self, x_bound, y_bound, z_bound):
        rows = [x_bound, y_bound, z_bound]
        self.voxel_size = torch.Tensor([r[2] for r in rows])
        self.voxel_coord = torch.Tensor([r[0] + r[2] / 2.0 for r in rows])
        self.voxel_num = torch.LongTensor([(r[1] - r[0]) / r[2] for r in rows])

    @property
    def lower(self):
        return (self.voxel_coord - self.voxel_size / 2.0).tolist()


def make_frustum(final_dim, downsample_factor, d_bound):
    H, W = final_dim
    fH, fW = H // downsample_factor, W // downsample_factor
    depth = torch.arange(*d_bound, dtype=torch.float)
    out = torch.ones(depth.numel(), fH, fW, 4)
    out[..., 0] = torch.linspace(0, W - 1, fW, dtype=torch.float)[None, None, :]
    out[..., 1] = torch.linspace(0, H - 1, fH, dtype=torch.float)[None, :, None]
    out[..., 2] = depth[:, None, None]
    return out


def stack_img_metas(img_metas, num_cams=4):
    """list[B][T] of dicts -> dict of (B,T,ncam,4,4) tensors (+ lidar2img / ida of the key frame).
    sensor2ego is the transpose of `currlidar2keycam`, exactly as the reference builds it."""
    K, A, S = [], [], []
    for sample in img_metas:
        k_t, a_t, s_t = [], [], []
        for m in sample:
            k = torch.zeros(num_cams, 4, 4)
            k[:, :3, :3] = torch.as_tensor(m["cam_intrinsic"], dtype=torch.float32)
            k[:, 3, 3] = 1.0
            k_t.append(k)
            a_t.append(torch.as_tensor(m["ida_mats"], dtype=torch.float32))
            s_t.append(torch.as_tensor(m["currlidar2keycam"], dtype=torch.float32).transpose(1, 2))
        K.append(torch.stack(k_t))
        A.append(torch.stack(a_t))
        S.append(torch.stack(s_t))
    out = {"intrin_mats": torch.stack(K), "ida_mats": torch.stack(A), "sensor2ego_mats": torch.stack(S)}
    out["lidar2img"] = torch.stack([torch.as_tensor(s[-1]["lidar2img"], dtype=torch.float32)
                                    for s in img_metas])
    out["ida_mat"] = out["ida_mats"][:, -1].clone()
    return out


def geometry_matrices(mats, sweep_index=-1):
    """(B*ncam, 2, 4, 4): [inv(ida), sensor2ego @ inv(intrin)] for one sweep."""
    ida = mats["ida_mats"][:, sweep_index]
    s2e = mats["sensor2ego_mats"][:, sweep_index]
    intrin = mats["intrin_mats"][:, sweep_index]
    pair = torch.stack([ida.inverse(), s2e.matmul(torch.inverse(intrin))], 2)
    return pair.reshape(-1, 2, 4, 4).contiguous()


def depth_mlp_input(mats):
    """(B*ncam, 22) camera-aware DepthNet input of the KEY frame (backbones/lss.py:206-231):
    [fx, fy, cx, cy, ida00, ida01, ida03, ida10, ida11, ida13, sensor2ego[:3,:] (12)]."""
    k = mats["intrin_mats"][:, -1]
    a = mats["ida_mats"][:, -1]
    s = mats["sensor2ego_mats"][:, -1][..., :3, :]
    B, N = k.shape[:2]
    head = torch.stack([k[..., 0, 0], k[..., 1, 1], k[..., 0, 2], k[..., 1, 2], a[..., 0, 0], a[..., 0, 1],
                        a[..., 0, 3], a[..., 1, 0], a[..., 1, 1], a[..., 1, 3]], -1)
    return torch.cat([head, s.reshape(B, N, 12)], -1).reshape(B * N, 22).contiguous()
