"""Model hyper-parameters of the reference config (own restatement of the `model=` section of
open_loop_training/configs/thinktwice.py:38-198; cited per field).  Pure data."""
import copy

POINT_CLOUD_RANGE = [-8.0, -19.2, -4.0, 30.4, 19.2, 10.0]        # CFG:40

CFG = dict(
    pred_len=4,                                                  # CFG:43
    turn_KP=0.75, turn_KI=0.75, turn_KD=0.3, turn_n=40,          # CFG:44-47
    speed_KP=5.0, speed_KI=0.5, speed_KD=1.0, speed_n=40,        # CFG:48-51
    brake_speed=0.4, brake_ratio=1.1, clip_delta=0.25,           # CFG:52-54
    aim_dist=4.0, angle_thresh=0.3, dist_thresh=10,              # CFG:55-57
    refine_num=5,                                                # CFG:70
    FPN_out_channels=[256, 256, 256, 256],                       # CFG:74
    point_cloud_range=POINT_CLOUD_RANGE,
    img_size=(448, 896),                                         # CFG:113,120
    num_cams=4, queue_length=2,                                  # CFG:99-104
    num_seg_type=11,                                             # CFG:108-109: 9 labels + 2
    value_weight=0.001, features_weight=0.05,                    # CFG:59-60 (training losses)
    use_depth=True, use_seg=True,                                # CFG:106-107 (training supervision)
)

MODEL = dict(
    type="EncoderDecoder",
    num_cams=4,
    decoder=dict(type="ThinkTwiceDecoder", bev_h=21, bev_w=21),  # CFG:124-129
    img_encoder=dict(
        type="LSS",
        x_bound=[-8.0, 30.4, 1.8285], y_bound=[-19.2, 19.2, 1.8285],   # CFG:133-134
        z_bound=[-4, 10, 14], d_bound=[1.0, 41.0, 0.5],                 # CFG:135-136
        final_dim=(448, 896), output_channels=256, downsample_factor=16, queue_len=2,  # CFG:137-140
        img_backbone_conf=dict(type="ResNet", depth=50, out_indices=[0, 1, 2, 3]),      # CFG:141-148
        img_neck_conf=dict(type="PAFPN", in_channels=[256, 512, 1024, 2048], num_outs=4,
                           out_channels=256),                                           # CFG:149-154
        depth_net_conf=dict(in_channels=512, mid_channels=512),                        # CFG:155
        seg_net_conf=dict(in_channels=512, out_channels=12),                           # CFG:156
        fpn_in_channels=[256, 256, 256, 256],                                          # CFG:157
    ),
    lidar_encoder=dict(
        type="LidarNet",
        pts_voxel_layer=dict(max_num_points=10, voxel_size=[0.0571428, 0.0571428, 0.2],
                             max_voxels=(120000, 160000), point_cloud_range=POINT_CLOUD_RANGE),  # CFG:161-165
        pts_voxel_encoder=dict(type="HardSimpleVFE", num_features=5),                            # CFG:166
        pts_middle_encoder=dict(
            type="SparseEncoder_fp32", in_channels=5, sparse_shape=[41, 672, 672], output_channels=128,
            base_channels=16,
            encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
            encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0))),                 # CFG:167-176
        pts_backbone=dict(type="SECOND", in_channels=256, out_channels=[128, 256], layer_nums=[5, 5],
                          layer_strides=[1, 2], bn_eps=1e-3),                                    # CFG:177-184
        pts_neck=dict(type="SECONDFPN", in_channels=[128, 256], out_channels=[256, 256],
                      upsample_strides=[1, 2], bn_eps=1e-3),                                     # CFG:185-192
    ),
)


def model_config(**overrides):
    """Deep copy of MODEL/CFG merged into one dict; `overrides` patch top-level image geometry for
    reduced-size parity cases (e.g. final_dim=(128, 256))."""
    m = copy.deepcopy(MODEL)
    c = copy.deepcopy(CFG)
    if "final_dim" in overrides:
        m["img_encoder"]["final_dim"] = tuple(overrides["final_dim"])
        c["img_size"] = tuple(overrides["final_dim"])
    if "sparse_shape" in overrides:
        m["lidar_encoder"]["pts_middle_encoder"]["sparse_shape"] = list(overrides["sparse_shape"])
    if "lidar_voxel_size" in overrides:
        m["lidar_encoder"]["pts_voxel_layer"]["voxel_size"] = list(overrides["lidar_voxel_size"])
    if "refine_num" in overrides:
        c["refine_num"] = int(overrides["refine_num"])
    m["cfg"] = c
    return m
