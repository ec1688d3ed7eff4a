"""Prepared layers: checkpoint tensors -> kernel-layout weights + fused epilogue parameters.

A `Conv` is one launch of tt_conv2d_fwd (implicit GEMM on MFMA) with BN/bias/activation/residual
fused; `Rows` are the same kernel used as a row-batched nn.Linear (always f32: tiny M).
"""
import torch

from . import _lib, ops, weights

ACT = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "sigmoid": _lib.ACT_SIGMOID, "gelu": _lib.ACT_GELU,
       "softplus": _lib.ACT_SOFTPLUS}


class Conv:
    def __init__(self, w, scale, shift, stride=1, pad=0, dil=1, act="none", pixel_shuffle2=False, x3=False):
        self.w, self.scale, self.shift = w, scale, shift
        # precision mode "f32x3": the same weights pre-split into bf16 (hi, lo) pairs ride along; the C side uses them
        # wherever the layer fits the LDS-DMA kernel and the exact f32 path on `w` elsewhere
        self.w_x3 = weights.split_pairs_x3(w) if x3 else None
        self.stride, self.pad, self.dil = stride, pad, dil
        self.act = ACT[act]
        self.ps2 = pixel_shuffle2
        self.cin = w.shape[-1]

    def __call__(self, x, **kw):
        return ops.conv2d(x, self.w, stride=self.stride, pad=self.pad, dil=self.dil, scale=self.scale,
                          shift=self.shift, act=self.act, pixel_shuffle2=self.ps2, w_x3=self.w_x3, **kw)


def _dev(t, device):
    return None if t is None else t.to(device=device, dtype=torch.float32).contiguous()


def conv_from_sd(sd, name, dtype, device, bn=None, eps=1e-5, stride=1, pad=0, dil=1, act="none",
                 cin_pad=None, weight=None, cin_lo=0):
    """nn.Conv2d `name` (+ optional eval BatchNorm `bn`) -> Conv.  `weight` (+ `cin_lo`): use this input-channel slice
    [cin_lo, cin_lo + weight.shape[1]) of the layer's weight instead of the whole tensor."""
    w = sd[name + ".weight"] if weight is None else weight
    bias = sd.get(name + ".bias")
    wq = weights.prep_conv_weight(w.to(device), dtype, cin_pad)
    if bn is not None:
        scale, shift = weights.fold_bn(sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                                       sd[bn + ".running_var"], eps, bias)
    else:
        scale, shift = None, bias
    # what the training tape needs for this layer's parameter gradients (autodiff.py)
    from . import autodiff
    full = sd[name + ".weight"]
    autodiff.CONV_META[id(wq)] = autodiff.ConvMeta(
        name, w.shape[1], bn,
        None if bn is None else _dev(sd[bn + ".running_mean"], device),
        None if bn is None else torch.sqrt(_dev(sd[bn + ".running_var"], device) + eps), _dev(bias, device),
        kind="conv" if weight is None else "cin_slice", full_shape=tuple(full.shape), lo=cin_lo)
    return Conv(wq, _dev(scale, device), _dev(shift, device), stride, pad, dil, act, x3=dtype == weights.X3)


def conv_from_weight(w, dtype, scale=None, shift=None, **kw):
    """Kernel-layout weight tensor [Cout,KH,KW,Cin_p] (storage dtype) -> Conv in the given precision mode."""
    return Conv(w, scale, shift, x3=dtype == weights.X3, **kw)


def linear_from_sd(sd, name, device, act="none", in_pad=None, dtype=torch.float32, bn=None, eps=1e-5):
    """nn.Linear `name` -> Conv over rows ([R,1,1,in] input)."""
    w = sd[name + ".weight"]
    wq = weights.prep_linear_weight(w.to(device), dtype, in_pad)
    bias = sd.get(name + ".bias")
    if bn is not None:
        scale, shift = weights.fold_bn(sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                                       sd[bn + ".running_var"], eps, bias)
    else:
        scale, shift = None, bias
    from . import autodiff
    autodiff.CONV_META[id(wq)] = autodiff.ConvMeta(
        name, w.shape[1], bn,
        None if bn is None else _dev(sd[bn + ".running_mean"], device),
        None if bn is None else torch.sqrt(_dev(sd[bn + ".running_var"], device) + eps), _dev(bias, device),
        kind="linear")
    return Conv(wq, _dev(scale, device), _dev(shift, device), act=act, x3=dtype == weights.X3)


def deconv2x2_from_sd(sd, name, dtype, device, bn=None, eps=1e-5, act="none"):
    """nn.ConvTranspose2d(k=2, s=2) -> 1x1 GEMM to 4*Cout + pixel shuffle in the epilogue."""
    wq = weights.prep_deconv2x2_weight(sd[name + ".weight"].to(device), dtype)
    bias = sd.get(name + ".bias")
    from . import autodiff
    autodiff.CONV_META[id(wq)] = autodiff.ConvMeta(
        name, sd[name + ".weight"].shape[0], bn,
        None if bn is None else _dev(sd[bn + ".running_mean"], device),
        None if bn is None else torch.sqrt(_dev(sd[bn + ".running_var"], device) + eps), _dev(bias, device))
    if bn is not None:
        scale, shift = weights.fold_bn(sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                                       sd[bn + ".running_var"], eps, bias)
    else:
        scale, shift = None, bias
    return Conv(wq, _dev(scale, device), _dev(shift, device), act=act, pixel_shuffle2=True, x3=dtype == weights.X3)


def rows(x):
    """(R, C) -> (R,1,1,C) view for the row-batched linear path."""
    return x.view(x.shape[0], 1, 1, x.shape[1])


def unrows(x):
    return x.view(x.shape[0], x.shape[-1])


def bn_affine(sd, bn, device, eps=1e-5):
    s, t = weights.fold_bn(sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                           sd[bn + ".running_var"], eps)
    s, t = _dev(s, device), _dev(t, device)
    from . import autodiff
    autodiff.AFFINE_META[id(s)] = (bn, _dev(sd[bn + ".running_mean"], device),
                                   torch.sqrt(_dev(sd[bn + ".running_var"], device) + eps))
    return s, t
