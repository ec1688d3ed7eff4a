"""ctypes binding of oracle/liboracle_ref.so (plain-C restatement; test infrastructure)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_ref.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def voxel_pool_fwd(geom_xyz, feats, voxel_num, acc64=True, want_pos_memo=True):
    """geom_xyz int32 [B,Np,3], feats f32 [B,Np,C] (numpy) -> out [B,Y,X,C], pos_memo [B,Np,3]."""
    geom_xyz = np.ascontiguousarray(geom_xyz, dtype=np.int32)
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    B, Np, C = feats.shape
    vx, vy, vz = (int(v) for v in voxel_num)
    out = np.zeros((B, vy, vx, C), dtype=np.float32)
    memo = np.full((B, Np, 3), -1, dtype=np.int32) if want_pos_memo else None
    rc = lib().oracle_voxel_pool_fwd(B, Np, C, vx, vy, vz, _p(geom_xyz), _p(feats), _p(out),
                                     _p(memo) if memo is not None else None, int(bool(acc64)))
    assert rc == 1
    return out, memo


def voxel_pool_bwd(pos_memo, grad_out_bhwc):
    pos_memo = np.ascontiguousarray(pos_memo, dtype=np.int32)
    g = np.ascontiguousarray(grad_out_bhwc, dtype=np.float32)
    B, Np, _ = pos_memo.shape
    _, vy, vx, C = g.shape
    grad_in = np.empty((B, Np, C), dtype=np.float32)
    rc = lib().oracle_voxel_pool_bwd(B, Np, C, vx, vy, _p(pos_memo), _p(g), _p(grad_in))
    assert rc == 1
    return grad_in
