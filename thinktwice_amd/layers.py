"""Prepared layers: checkpoint tensors -> kernel-layout weights + fused epilogue parameters.

A `Conv` is one launch of tt_conv2d_fwd (implicit GEMM on MFMA) with BN/bias/activation/residual
fused; `Rows` are the same kernel used as a row-batched nn.Linear (always f32: tiny M).
"""
import torch

from . import _lib, ops, weights

ACT = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "sigmoid": _lib.ACT_SIGMOID, "gelu": _lib.ACT_GELU,
       "softplus": _lib.ACT_SOFTPLUS}


# Train-mode switches (EncoderDecoder.train() / forward_train set them; thinktwice_amd/ops.py::batchnorm_train does the work):
BN_TRAIN = False     # BatchNorm layers normalise with batch statistics and update their running statistics (model.train())
BN_GROUPS = 1        # equal row groups of the activations in flight, each with its own statistics (camera trunk: T sweeps)


class bn_groups:
    """`with layers.bn_groups(T):` -- the activations inside are T equal image groups (one per sweep)."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        global BN_GROUPS
        self.saved, BN_GROUPS = BN_GROUPS, self.n

    def __exit__(self, *exc):
        global BN_GROUPS
        BN_GROUPS = self.saved
        return False


def _bn_spec(sd, bn, device, eps, momentum):
    return ops.BNSpec(bn, _dev(sd[bn + ".weight"], device), _dev(sd[bn + ".bias"], device),
                      _dev(sd[bn + ".running_mean"], device), _dev(sd[bn + ".running_var"], device), eps, momentum)


class Conv:
    def __init__(self, w, scale, shift, stride=1, pad=0, dil=1, act="none", pixel_shuffle2=False, x3=False, bn=None,
                 bias=None, h2=False):
        self.w, self.scale, self.shift = w, scale, shift
        self.bn, self.bias = bn, bias            # train mode: ops.BNSpec + the conv's own bias (device f32) or None
        # precision mode "f32x3": the same weights pre-split into bf16 (hi, lo) pairs ride along; the C side uses them
        # wherever the layer fits the LDS-DMA kernel and the exact f32 path on `w` elsewhere
        self.w_x3 = weights.split_pairs_x3(w) if x3 else None
        # layer mode "h2" (half-storage stage of the mixed mode): `w` arrives as the PREPARED f32 weights; the kernel reads the
        # f16 (hi, lo) pair, `w` itself stays as the half-typed shape carrier the C ABI asks for
        self.w_h2 = None
        if h2:
            self.w_h2 = weights.split_pairs_h2(w)
            self.w = w.to(torch.float16)
        self.stride, self.pad, self.dil = stride, pad, dil
        self.act = ACT[act]
        self.ps2 = pixel_shuffle2
        self.cin = w.shape[-1]

    def __call__(self, x, **kw):
        if BN_TRAIN and self.bn is not None:
            return self._bn_train(x, **kw)
        return ops.conv2d(x, self.w, stride=self.stride, pad=self.pad, dil=self.dil, scale=self.scale,
                          shift=self.shift, act=self.act, pixel_shuffle2=self.ps2, w_x3=self.w_x3, w_h2=self.w_h2, **kw)

    def _bn_train(self, x, res1=None, res1_coff=0, res2=None, res2_coff=0, out=None, out_coff=0, out_dtype=None, **kw):
        """model.train(): raw convolution (+ bias) into a dense buffer, then batch-statistics BatchNorm + residuals +
        activation into the caller's output window (ops.batchnorm_train).  A per-image shift (`shift_n`) must be the
        UNSCALED one (the caller's business: lss.py)."""
        z = ops.conv2d(x, self.w, stride=self.stride, pad=self.pad, dil=self.dil, scale=None, shift=self.bias, act=0,
                       pixel_shuffle2=self.ps2, w_x3=self.w_x3, out_dtype=torch.float32, bn_raw=True, **kw)
        return ops.batchnorm_train(z, self.bn, self.act, res1=res1, res1_coff=res1_coff, res2=res2, res2_coff=res2_coff,
                                   out=out, out_coff=out_coff, groups=BN_GROUPS)


def _dev(t, device):
    return None if t is None else t.to(device=device, dtype=torch.float32).contiguous()


def conv_from_sd(sd, name, dtype, device, bn=None, eps=1e-5, stride=1, pad=0, dil=1, act="none",
                 cin_pad=None, weight=None, cin_lo=0, bn_momentum=0.1):
    """nn.Conv2d `name` (+ optional eval BatchNorm `bn`) -> Conv.  `weight` (+ `cin_lo`): use this input-channel slice
    [cin_lo, cin_lo + weight.shape[1]) of the layer's weight instead of the whole tensor."""
    w = sd[name + ".weight"] if weight is None else weight
    bias = sd.get(name + ".bias")
    h2 = isinstance(dtype, str) and dtype == weights.H2
    wq = weights.prep_conv_weight(w.to(device), torch.float32 if h2 else dtype, cin_pad)
    if bn is not None:
        scale, shift = weights.fold_bn(sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                                       sd[bn + ".running_var"], eps, bias)
    else:
        scale, shift = None, bias
    # what the training tape needs for this layer's parameter gradients (autodiff.py)
    from . import autodiff
    full = sd[name + ".weight"]
    autodiff.CONV_META[wq] = autodiff.ConvMeta(
        name, w.shape[1], bn,
        None if bn is None else _dev(sd[bn + ".running_mean"], device),
        None if bn is None else torch.sqrt(_dev(sd[bn + ".running_var"], device) + eps), _dev(bias, device),
        kind="conv" if weight is None else "cin_slice", full_shape=tuple(full.shape), lo=cin_lo)
    if bn is not None:
        autodiff.CONV_META[wq].scale_ref = _dev(scale, device)
    return Conv(wq, _dev(scale, device), _dev(shift, device), stride, pad, dil, act, x3=weights.is_x3(dtype),
                bn=None if bn is None else _bn_spec(sd, bn, device, eps, bn_momentum), bias=_dev(bias, device), h2=h2)


def conv_from_weight(w, dtype, scale=None, shift=None, **kw):
    """Kernel-layout weight tensor [Cout,KH,KW,Cin_p] (storage dtype) -> Conv in the given precision mode."""
    return Conv(w, scale, shift, x3=weights.is_x3(dtype), **kw)


def linear_from_sd(sd, name, device, act="none", in_pad=None, dtype=torch.float32, bn=None, eps=1e-5, bn_momentum=0.1):
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
    autodiff.CONV_META[wq] = autodiff.ConvMeta(
        name, w.shape[1], bn,
        None if bn is None else _dev(sd[bn + ".running_mean"], device),
        None if bn is None else torch.sqrt(_dev(sd[bn + ".running_var"], device) + eps), _dev(bias, device),
        kind="linear")
    return Conv(wq, _dev(scale, device), _dev(shift, device), act=act, x3=weights.is_x3(dtype),
                bn=None if bn is None else _bn_spec(sd, bn, device, eps, bn_momentum), bias=_dev(bias, device))


def deconv2x2_from_sd(sd, name, dtype, device, bn=None, eps=1e-5, act="none", bn_momentum=0.1):
    """nn.ConvTranspose2d(k=2, s=2) -> 1x1 GEMM to 4*Cout + pixel shuffle in the epilogue."""
    wq = weights.prep_deconv2x2_weight(sd[name + ".weight"].to(device), dtype)
    bias = sd.get(name + ".bias")
    from . import autodiff
    autodiff.CONV_META[wq] = autodiff.ConvMeta(
        name, sd[name + ".weight"].shape[0], bn,
        None if bn is None else _dev(sd[bn + ".running_mean"], device),
        None if bn is None else torch.sqrt(_dev(sd[bn + ".running_var"], device) + eps), _dev(bias, device))
    if bn is not None:
        scale, shift = weights.fold_bn(sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                                       sd[bn + ".running_var"], eps, bias)
    else:
        scale, shift = None, bias
    return Conv(wq, _dev(scale, device), _dev(shift, device), act=act, pixel_shuffle2=True, x3=weights.is_x3(dtype),
                bn=None if bn is None else _bn_spec(sd, bn, device, eps, bn_momentum), bias=_dev(bias, device))


def rows(x):
    """(R, C) -> (R,1,1,C) view for the row-batched linear path."""
    return x.view(x.shape[0], 1, 1, x.shape[1])


def unrows(x):
    return x.view(x.shape[0], x.shape[-1])


class BNAffine:
    """A stand-alone BatchNorm over rows (BatchNorm1d on a feature vector): eval = folded affine (`[0]` scale, `[1]` shift,
    like the tuple this used to be), train = batch statistics."""

    def __init__(self, scale, shift, spec):
        self.scale, self.shift, self.spec = scale, shift, spec

    def __getitem__(self, i):
        return (self.scale, self.shift)[i]

    def __call__(self, x, act=0, out=None, running_updates=1):
        """x (R, C) f32 rows -> (R, C') rows of `out` (allocated if None; wider outputs keep their extra columns).
        `running_updates` = n (train mode): the running statistics take n momentum updates with this batch's statistics
        (the reference runs this BatchNorm once per sweep on the SAME input: n updates of momentum m are one update of
        momentum 1 - (1 - m)^n)."""
        if BN_TRAIN:
            if running_updates == 1:
                return ops.batchnorm_train(x.contiguous(), self.spec, act, out=out)
            import copy
            spec = copy.copy(self.spec)               # shares gamma / beta / running buffers; only the momentum differs
            spec.momentum = 1.0 - (1.0 - float(self.spec.momentum)) ** int(running_updates)
            return ops.batchnorm_train(x.contiguous(), spec, act, out=out)
        return ops.affine_rows(x, self.scale, self.shift, act=act, out=out)


def bn_affine(sd, bn, device, eps=1e-5, momentum=0.1):
    s, t = weights.fold_bn(sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                           sd[bn + ".running_var"], eps)
    s, t = _dev(s, device), _dev(t, device)
    from . import autodiff
    autodiff.AFFINE_META[s] = (bn, _dev(sd[bn + ".running_mean"], device),
                                   torch.sqrt(_dev(sd[bn + ".running_var"], device) + eps))
    return BNAffine(s, t, _bn_spec(sd, bn, device, eps, momentum))
