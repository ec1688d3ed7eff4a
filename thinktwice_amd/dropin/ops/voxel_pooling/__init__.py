"""Drop-in for the reference package `open_loop_training/ops/voxel_pooling/__init__.py:1-3`
(`from .voxel_pooling import voxel_pooling`; `__all__ = ['voxel_pooling']`): the same name from the same import path,
backed by libthinktwice_hip.so (tt_voxel_pool_fwd_ws / tt_voxel_pool_bwd).  `backbones/lss.py:12`
(`from ops.voxel_pooling import voxel_pooling`) then binds the HIP operator without an edit.

Two ways to use it:
  * overlay: copy this directory's two files over `open_loop_training/ops/voxel_pooling/` -- `__init__.py` replaces the
    reference's and takes the whole operator (autograd Function included) from thinktwice_amd;
  * extension only: keep the reference's own `voxel_pooling.py` and copy just `voxel_pooling_ext.py` next to it -- its
    `from . import voxel_pooling_ext` then finds `voxel_pooling_forward_wrapper` (voxel_pooling_forward.cpp:24-37) here
    instead of in the compiled CUDA extension.
"""
from thinktwice_amd.voxel_pooling import voxel_pooling

__all__ = ['voxel_pooling']
