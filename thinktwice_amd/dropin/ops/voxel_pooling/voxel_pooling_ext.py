"""Drop-in for the compiled extension module `voxel_pooling_ext` that the reference's
`ops/voxel_pooling/voxel_pooling.py:5,42-53` imports and calls: one symbol, `voxel_pooling_forward_wrapper`, argument for
argument the pybind entry of `src/voxel_pooling_forward.cpp:24-37,66-69` (0-dim tensors are accepted for the six integer
arguments, as pybind does; `output_features` pre-zeroed and `pos_memo` pre-filled with -1 by the caller; returns 1)."""
from thinktwice_amd.voxel_pooling import voxel_pooling_forward_wrapper

__all__ = ['voxel_pooling_forward_wrapper']
