"""Oracle (TEST INFRASTRUCTURE): frustum, camera->ego geometry and voxel index of the
Lift-Splat encoder.  Restates
  LSS.create_frustum        backbones/lss.py:454-471
  LSS.get_geometry          backbones/lss.py:474-512   (bda_mat is None in this model)
  LSS.voxel_pooling_method  backbones/lss.py:629-631   (index math only)
  LSS.forward meta assembly backbones/lss.py:667-687,704-707
Pinned by tests/golden/f12_geometry.npz (generated from the reference's own methods).
"""
import torch


def create_frustum(final_dim=(448, 896), downsample=16, d_bound=(1.0, 41.0, 0.5)):
    ogfH, ogfW = final_dim
    fH, fW = ogfH // downsample, ogfW // downsample
    d = torch.arange(*d_bound, dtype=torch.float)
    D = d.shape[0]
    xs = torch.linspace(0, ogfW - 1, fW, dtype=torch.float)
    ys = torch.linspace(0, ogfH - 1, fH, dtype=torch.float)
    fr = torch.empty(D, fH, fW, 4)
    fr[..., 0] = xs.view(1, 1, fW)
    fr[..., 1] = ys.view(1, fH, 1)
    fr[..., 2] = d.view(D, 1, 1)
    fr[..., 3] = 1.0
    return fr


def assemble_camera_mats(img_metas):
    """-> intrin (B,T,4,4,4), ida (B,T,4,4,4), sensor2ego (B,T,4,4,4), lidar2img (B,4,4,4),
    current_ida (B,4,4,4).  NOTE sensor2ego is the TRANSPOSE of currlidar2keycam (lss.py:677)."""
    intr, ida, s2e = [], [], []
    for per_sample in img_metas:
        a, b, c = [], [], []
        for m in per_sample:
            k = torch.zeros(4, 4, 4)
            k[:, :3, :3] = torch.as_tensor(m["cam_intrinsic"])
            k[:, 3, 3] = 1
            a.append(k)
            b.append(torch.as_tensor(m["ida_mats"]))
            c.append(torch.as_tensor(m["currlidar2keycam"]).permute(0, 2, 1))
        intr.append(torch.stack(a))
        ida.append(torch.stack(b))
        s2e.append(torch.stack(c))
    intr, ida, s2e = torch.stack(intr), torch.stack(ida), torch.stack(s2e)
    lidar2img = torch.stack([torch.as_tensor(s[-1]["lidar2img"]) for s in img_metas])
    return intr, ida, s2e, lidar2img, ida[:, -1].clone()


def geometry_mats(sensor2ego, intrin, ida):
    """The two per-camera 4x4 matrices get_geometry applies: inv(ida) and sensor2ego@inv(intrin)."""
    return ida.inverse(), sensor2ego.matmul(torch.inverse(intrin))


def _dot4_seq(m_row, p):
    # explicit k-ordered products and sums, each rounded to f32 (no FMA): the arithmetic the
    # HIP kernel performs; checked bit-identical (as voxel indices) with the reference's
    # matmul formulation in tests/golden/gen_golden.py::gen_f12.
    s = m_row[..., 0] * p[0]
    s = s + m_row[..., 1] * p[1]
    s = s + m_row[..., 2] * p[2]
    s = s + m_row[..., 3] * p[3]
    return s


def get_geometry(frustum, sensor2ego, intrin, ida):
    """frustum (D,H,W,4); mats (B,N,4,4) -> ego xyz (B,N,D,H,W,3)."""
    inv_ida, comb = geometry_mats(sensor2ego, intrin, ida)
    B, N = sensor2ego.shape[:2]
    f = [frustum[..., i].view(1, 1, *frustum.shape[:3]) for i in range(4)]
    M = inv_ida.view(B, N, 1, 1, 1, 4, 4)
    p = [_dot4_seq(M[..., r, :], f) for r in range(4)]
    p[0] = p[0] * p[2]
    p[1] = p[1] * p[2]
    Cm = comb.view(B, N, 1, 1, 1, 4, 4)
    g = [_dot4_seq(Cm[..., r, :], p) for r in range(3)]
    return torch.stack(g, -1)


def voxel_index(geom, voxel_coord, voxel_size):
    """((geom - (voxel_coord - voxel_size/2)) / voxel_size).int()  -- truncation toward zero."""
    return ((geom - (voxel_coord - voxel_size / 2.0)) / voxel_size).int()


def voxel_constants(x_bound=(-8.0, 30.4, 1.8285), y_bound=(-19.2, 19.2, 1.8285), z_bound=(-4, 10, 14)):
    rows = [x_bound, y_bound, z_bound]
    voxel_size = torch.Tensor([r[2] for r in rows])
    voxel_coord = torch.Tensor([r[0] + r[2] / 2.0 for r in rows])
    voxel_num = torch.LongTensor([(r[1] - r[0]) / r[2] for r in rows])
    return voxel_size, voxel_coord, voxel_num
