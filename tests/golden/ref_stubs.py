"""Import shims that let the reference's OWN python modules run in the build container.

Runs ONLY where /root/reference exists (the build container); never on the GPU box.
It installs thin `sys.modules` stand-ins for the third-party packages the reference
imports but which are not installed here (mmcv, mmdet, mmdet3d, mmcls, spconv, cv2,
torchvision) so that the in-repo reference files

    open_loop_training/code/encoder_decoder_framework.py
    open_loop_training/code/model_code/backbones/lss.py
    open_loop_training/code/model_code/dense_heads/{thinktwice_decoder,multi_scale_deformable_attn_function,utils}.py
    open_loop_training/code/utils.py
    open_loop_training/ops/voxel_pooling/voxel_pooling.py

can be imported UNMODIFIED from where they lie and executed on CPU to produce golden
vectors (tests/golden/gen_golden.py).  No reference source is copied: the stand-ins only
provide (a) identity decorators / no-op registries (legitimate because fp16 is disabled in
the reference config, configs/thinktwice.py:202), (b) `BaseModule` = `nn.Module`, and
(c) for third-party ARITHMETIC (ResNet, PAFPN, BasicBlock, DCN, MSDA core, voxelization,
sparse encoder, SECOND, SECONDFPN) adapters onto this repo's oracle restatements -- those
parts are therefore NOT pinned by the reference ("parity unpinned").
"""
import importlib
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = "/root/reference"
OLT = os.path.join(REF_ROOT, "open_loop_training")
REF_PKG = "ttref_code"  # the reference package dir is literally called `code` (clashes with stdlib)


def reference_available():
    return os.path.isdir(OLT)


class _Anything:
    """Permissive placeholder: callable, attribute-able, usable as a decorator factory."""

    def __init__(self, name="stub"):
        self._name = name

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Anything(self._name + "()")

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Anything(self._name + "." + item)

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Anything(self.__name__ + "." + item)


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = _StubModule(name)
        m.__path__ = []  # behave like a package so `import a.b.c` works
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(_mod(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.module_dict[cls.__name__] = cls
            return cls
        return deco


def _identity_decorator_factory(*a, **k):
    def deco(fn):
        return fn
    return deco


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    # mmcv.cnn.xavier_init semantics: silently does nothing for objects without .weight
    if hasattr(module, "weight") and module.weight is not None:
        if distribution == "uniform":
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


_installed = False
# adapters for third-party arithmetic; filled by install(third_party=...)
THIRD_PARTY = {}


def install(third_party=None):
    """Install the stand-ins. `third_party` maps names to callables/classes:
    'msda_core', 'BasicBlock', 'build_conv_layer', 'build_backbone', 'build_neck',
    'build_head', 'voxel_pooling_ext_fwd'."""
    global _installed
    if third_party:
        THIRD_PARTY.update(third_party)
    if _installed:
        return
    _installed = True
    if OLT not in sys.path:
        sys.path.insert(0, OLT)

    def tp(name):
        def call(*a, **k):
            if name not in THIRD_PARTY:
                raise RuntimeError(f"third-party adapter '{name}' not provided to ref_stubs.install")
            return THIRD_PARTY[name](*a, **k)
        return call

    class _LazyBasicBlock:
        def __new__(cls, *a, **k):
            return THIRD_PARTY["BasicBlock"](*a, **k)

    _mod("cv2")
    _mod("mmcv")
    _mod("mmcv.runner", BaseModule=BaseModule, force_fp32=_identity_decorator_factory,
         auto_fp16=_identity_decorator_factory)
    _mod("mmcv.runner.base_module", BaseModule=BaseModule, ModuleList=nn.ModuleList,
         Sequential=nn.Sequential)
    _mod("mmcv.cnn", xavier_init=xavier_init, constant_init=constant_init,
         build_conv_layer=lambda cfg=None, *a, **k: tp("build_conv_layer")(cfg, *a, **k))
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.registry", ATTENTION=_Registry("attention"),
         TRANSFORMER_LAYER_SEQUENCE=_Registry("tls"))
    _mod("mmcv.cnn.bricks.transformer", TransformerLayerSequence=nn.Module)

    class _ExtLoader:
        @staticmethod
        def load_ext(*a, **k):
            return None
    _mod("mmcv.utils", ext_loader=_ExtLoader, ConfigDict=ConfigDict,
         build_from_cfg=_Anything("build_from_cfg"),
         deprecated_api_warning=_identity_decorator_factory, to_2tuple=lambda x: (x, x))
    _mod("mmcv.ops")
    _mod("mmcv.ops.multi_scale_deform_attn",
         multi_scale_deformable_attn_pytorch=lambda *a, **k: tp("msda_core")(*a, **k))
    _mod("mmdet")
    _mod("mmdet.core", multi_apply=_Anything("multi_apply"), reduce_mean=_Anything("reduce_mean"))
    _mod("mmdet.models", DETECTORS=_Registry("det"), BACKBONES=_Registry("bb"),
         HEADS=_Registry("heads"), NECKS=_Registry("necks"),
         build_backbone=lambda cfg: tp("build_backbone")(cfg))
    _mod("mmdet.models.backbones")
    _mod("mmdet.models.backbones.resnet", BasicBlock=_LazyBasicBlock)
    _mod("mmdet.models.necks")
    _mod("mmdet.models.necks.pafpn", PAFPN=nn.Module)
    _mod("mmcls")
    _mod("mmcls.models")
    builder = _mod("mmdet3d.models.builder", MIDDLE_ENCODERS=_Registry("me"),
                   build_backbone=lambda cfg: tp("build_backbone")(cfg),
                   build_head=lambda cfg: tp("build_head")(cfg))
    _mod("mmdet3d")
    _mod("mmdet3d.models", builder=builder, build_neck=lambda cfg: tp("build_neck")(cfg))
    _mod("torchvision")

    class _InterpolationMode:          # [3P] torchvision.transforms.InterpolationMode
        NEAREST = "nearest"
        BILINEAR = "bilinear"

    class _Resize:                     # [3P] torchvision.transforms.Resize on float TENSORS = F.interpolate (torchvision 0.13:
        # default interpolation bilinear, align_corners=False, antialias off for tensors)
        def __init__(self, size, interpolation="bilinear"):
            assert interpolation in ("nearest", "bilinear")
            self.size = tuple(size)
            self.mode = interpolation

        def __call__(self, x):
            import torch.nn.functional as F
            lead = x.shape[:-3]
            y = x.reshape(-1, *x.shape[-3:]).float()
            if self.mode == "nearest":
                y = F.interpolate(y, size=self.size, mode="nearest")
            else:
                y = F.interpolate(y, size=self.size, mode="bilinear", align_corners=False)
            return y.reshape(*lead, *y.shape[-3:]).to(x.dtype)

    class _Normalize:                  # [3P] torchvision.transforms.Normalize: (x - mean) / std per channel
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
            self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean.to(x.dtype)) / self.std.to(x.dtype)

    class _Compose:                    # [3P] torchvision.transforms.Compose
        def __init__(self, ts):
            self.ts = list(ts)

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    _mod("torchvision.transforms", Resize=_Resize, InterpolationMode=_InterpolationMode, Normalize=_Normalize,
         Compose=_Compose)

    # the reference's python wrapper imports `voxel_pooling_ext` relative to ops.voxel_pooling
    ext = types.ModuleType("ops.voxel_pooling.voxel_pooling_ext")
    ext.voxel_pooling_forward_wrapper = lambda *a: tp("voxel_pooling_ext_fwd")(*a)
    sys.modules["ops.voxel_pooling.voxel_pooling_ext"] = ext

    # synthetic parent package for open_loop_training/code (avoids running its __init__)
    pkg = types.ModuleType(REF_PKG)
    pkg.__path__ = [os.path.join(OLT, "code")]
    sys.modules[REF_PKG] = pkg
    for sub in ("model_code", "model_code.backbones", "model_code.dense_heads"):
        m = types.ModuleType(f"{REF_PKG}.{sub}")
        m.__path__ = [os.path.join(OLT, "code", *sub.split("."))]
        sys.modules[f"{REF_PKG}.{sub}"] = m


def ref_import(dotted):
    """Import a reference module: 'utils', 'encoder_decoder_framework',
    'model_code.backbones.lss', 'model_code.dense_heads.thinktwice_decoder', ...
    or 'ops.voxel_pooling.voxel_pooling'."""
    install()
    if dotted.startswith("ops."):
        return importlib.import_module(dotted)
    return importlib.import_module(f"{REF_PKG}.{dotted}")
