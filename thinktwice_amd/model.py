"""Convenience constructors: build the registered `EncoderDecoder` from the restated config
(the reference does `build_model(cfg.model)`, train.py:221-224 / thinktwice_agent.py:168)."""
import torch

from . import config as _config
from . import decoder, encoder_decoder, lidarnet, lss  # noqa: F401  (register modules)
from .registry import build_model


def build_thinktwice(dtype=torch.float32, device="cuda", lidar_dtype=None, **overrides):
    cfg = _config.model_config(**overrides)
    model = build_model(dict(type="EncoderDecoder", img_encoder=cfg["img_encoder"], decoder=cfg["decoder"],
                             lidar_encoder=cfg["lidar_encoder"], num_cams=cfg["num_cams"], train_cfg=cfg["cfg"],
                             test_cfg=cfg["cfg"]), dtype=dtype, device=device, lidar_dtype=lidar_dtype)
    return model, cfg


def batch_to_device(batch, device="cuda"):
    out = dict(batch)
    for k in ("img", "points", "speed", "target_point", "target_command", "target_command_raw"):
        if k in out and torch.is_tensor(out[k]):
            out[k] = out[k].to(device)
    return out
