"""Import-path drop-ins: files laid out like the reference tree, to be overlaid on it.

    thinktwice_amd/dropin/ops/voxel_pooling/  ->  open_loop_training/ops/voxel_pooling/
"""
