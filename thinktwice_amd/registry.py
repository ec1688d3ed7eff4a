"""Minimal mmcv-compatible Registry surface (`register_module()`, `build(cfg)`), so the reference's
config dicts (`type='LSS'`, ...) construct this package's modules
(mmdet DETECTORS/BACKBONES/HEADS usage: encoder_decoder_framework.py:23,51,56,72)."""


class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **extra):
        cfg = dict(cfg)
        t = cfg.pop("type")
        cls = self.module_dict.get(t)
        if cls is None:
            raise KeyError(f"{t} is not in the {self.name} registry")
        cfg.update(extra)
        return cls(**cfg)


DETECTORS = Registry("detector")
BACKBONES = Registry("backbone")
HEADS = Registry("head")
NECKS = Registry("neck")
MIDDLE_ENCODERS = Registry("middle_encoder")


def build_backbone(cfg, **extra):
    return BACKBONES.build(cfg, **extra)


def build_neck(cfg, **extra):
    return NECKS.build(cfg, **extra)


def build_head(cfg, **extra):
    return HEADS.build(cfg, **extra)


def build_model(cfg, **extra):
    return DETECTORS.build(cfg, **extra)
