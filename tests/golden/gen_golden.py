"""Generate the committed golden fixtures by running the REFERENCE's own python modules.

Run in the build container only (needs /root/reference):
    python tests/golden/gen_golden.py [--only F3,...]

Each fixture is data: inputs (or the seeds that regenerate them) + the outputs the
reference code produced.  See tests/golden/README.md for the list.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_stubs  # noqa: E402


def _save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------
# F3: VoxelPooling python wrapper semantics (ops/voxel_pooling/voxel_pooling.py:10-55)
# ---------------------------------------------------------------------------
def gen_f3():
    from oracle import c_ref

    def ext_fwd(batch_size, num_points, num_channels, vx, vy, vz, geom, feats, out, memo):
        # stands in for the CUDA-only extension symbol (voxel_pooling_forward.cpp:24-37):
        # restated kernel semantics on the caller-allocated buffers
        o, m = c_ref.voxel_pool_fwd(geom.numpy(), feats.numpy(), (int(vx), int(vy), int(vz)))
        out += torch.from_numpy(o)
        sel = torch.from_numpy(m[..., 0] != -1)
        memo[sel] = torch.from_numpy(m)[sel]
        return 1

    ref_stubs.install({"voxel_pooling_ext_fwd": ext_fwd})
    vp = ref_stubs.ref_import("ops.voxel_pooling.voxel_pooling")

    g = torch.Generator().manual_seed(303)
    B, Np, C = 2, 2048, 8
    vx, vy, vz = 21, 21, 1
    geom = torch.stack([
        torch.randint(-3, vx + 3, (B, Np), generator=g),
        torch.randint(-3, vy + 3, (B, Np), generator=g),
        torch.randint(-1, vz + 1, (B, Np), generator=g),
    ], -1).to(torch.int32)
    # the (-1,0) -> 0 truncation case of LSS.voxel_pooling_method (lss.py:630-631): produce some
    # indices through the reference's own float -> .int() expression
    fl = torch.tensor([-0.999, -0.5, -1e-7, 0.0, 0.999, 20.999, 21.0, -1.0, -1.0001])
    trunc = fl.int()
    geom[0, : len(fl), 0] = trunc
    geom[0, : len(fl), 1] = 3
    geom[0, : len(fl), 2] = 0
    # long runs into a single cell (contention case) and an all-out-of-range tail
    geom[1, 100:400] = torch.tensor([5, 7, 0], dtype=torch.int32)
    geom[1, -64:] = torch.tensor([-1, -1, -1], dtype=torch.int32)
    feats = torch.randn(B, Np, C, generator=g)
    feats.requires_grad_(True)
    voxel_num = torch.tensor([vx, vy, vz])
    out = vp.voxel_pooling(geom.contiguous(), feats.contiguous(), voxel_num)  # [B,C,Y,X]
    grad_out = torch.randn(out.shape, generator=g)
    out.backward(grad_out)
    _save("f3_voxel_pool.npz", geom=geom.numpy(), feats=feats.detach().numpy(),
          voxel_num=np.array([vx, vy, vz]), out=out.detach().numpy(),
          trunc_float=fl.numpy(), trunc_int=trunc.numpy(),
          grad_out=grad_out.numpy(), grad_in=feats.grad.numpy())


# ---------------------------------------------------------------------------
# F12: camera-matrix assembly, frustum, geometry, voxel index
#      (backbones/lss.py:454-512, 629-631, 667-687)
# ---------------------------------------------------------------------------
def _lss_namespace(lss_mod):
    import types as _t
    from oracle import lss_geometry as og
    ns = _t.SimpleNamespace()
    ns.final_dim = (448, 896)
    ns.downsample_factor = 16
    ns.d_bound = [1.0, 41.0, 0.5]
    ns.frustum = lss_mod.LSS.create_frustum(ns)
    # buffers exactly as LSS.__init__ registers them (lss.py:386-397)
    x_bound, y_bound, z_bound = [-8.0, 30.4, 1.8285], [-19.2, 19.2, 1.8285], [-4, 10, 14]
    ns.voxel_size = torch.Tensor([row[2] for row in [x_bound, y_bound, z_bound]])
    ns.voxel_coord = torch.Tensor([row[0] + row[2] / 2.0 for row in [x_bound, y_bound, z_bound]])
    ns.voxel_num = torch.LongTensor([(row[1] - row[0]) / row[2] for row in [x_bound, y_bound, z_bound]])
    return ns


def gen_f12():
    from oracle import c_ref
    from oracle import lss_geometry as og
    from thinktwice_amd import synth
    captured = {}

    def ext_fwd(batch_size, num_points, num_channels, vx, vy, vz, geom, feats, out, memo):
        captured["geom_int"] = geom.clone()
        o, m = c_ref.voxel_pool_fwd(geom.numpy(), feats.numpy(), (int(vx), int(vy), int(vz)))
        out += torch.from_numpy(o)
        return 1

    ref_stubs.install({"voxel_pooling_ext_fwd": ext_fwd})
    lss = ref_stubs.ref_import("model_code.backbones.lss")
    ns = _lss_namespace(lss)
    # one non-identity curr2key for the previous sweep (dead data in the reference, A10 quirk)
    c2k = np.eye(4, dtype=np.float32)
    c2k[:2, :2] = [[np.cos(0.05), np.sin(0.05)], [-np.sin(0.05), np.cos(0.05)]]
    c2k[0, 3], c2k[1, 3] = 0.7, -0.2
    metas = synth.make_img_metas(2, curr2key=c2k)
    intr, ida, s2e, l2i, cur_ida = og.assemble_camera_mats(metas)
    # reference geometry for the key sweep (index -1), matmul formulation
    geom_ref = lss.LSS.get_geometry(ns, s2e[:, -1], intr[:, -1], ida[:, -1], None)
    B, N, D, H, W, _ = geom_ref.shape
    feats = torch.ones(B, N, D, H, W, 1)
    bev = lss.LSS.voxel_pooling_method(ns, geom_ref, feats.contiguous(), ns.voxel_num)
    idx_ref = captured["geom_int"].reshape(B, N, D, H, W, 3)
    # oracle restatement (explicit k-ordered arithmetic)
    geom_o = og.get_geometry(ns.frustum, s2e[:, -1], intr[:, -1], ida[:, -1])
    idx_o = og.voxel_index(geom_o, ns.voxel_coord, ns.voxel_size)
    mism = int((idx_o != idx_ref).any(-1).sum())
    print("geometry max abs diff", float((geom_o - geom_ref).abs().max()), "index mismatches", mism)
    hist = bev[:, 0].to(torch.int64)  # counts per cell [B,Y,X]
    inr = int(hist[0].sum())
    print("in-range points per sample:", inr, "max per cell", int(hist[0].max()))
    g = torch.Generator().manual_seed(12)
    sel = torch.randint(0, N * D * H * W, (256,), generator=g)
    _save("f12_geometry.npz",
          frustum_corners=ns.frustum[[0, 0, -1, -1], [0, -1, 0, -1], [0, -1, -1, 0]].numpy(),
          frustum_shape=np.array(ns.frustum.shape),
          intrin=intr.numpy(), ida=ida.numpy(), sensor2ego=s2e.numpy(), lidar2img=l2i.numpy(),
          cur_ida=cur_ida.numpy(), curr2key=c2k,
          voxel_size=ns.voxel_size.numpy(), voxel_coord=ns.voxel_coord.numpy(),
          voxel_num=ns.voxel_num.numpy(),
          sample_index=sel.numpy(),
          geom_sample=geom_ref.reshape(B, -1, 3)[:, sel].numpy(),
          idx_sample=idx_ref.reshape(B, -1, 3)[:, sel].numpy(),
          cell_hist=hist.numpy().astype(np.int32),
          idx_ref_packed=np.packbits(((idx_ref[0] >= 0) & (idx_ref[0] < ns.voxel_num.int())).all(-1).numpy().reshape(-1)),
          idx_mismatch_vs_oracle=np.array([mism]))


FIXTURES = {"F3": gen_f3, "F12": gen_f12}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    if not ref_stubs.reference_available():
        raise SystemExit("needs /root/reference (build container only)")
    names = [n for n in args.only.split(",") if n] or list(FIXTURES)
    for n in names:
        print(f"== {n}")
        FIXTURES[n]()


if __name__ == "__main__":
    main()
