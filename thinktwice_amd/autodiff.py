"""Reverse-mode tape for the training step: the backward pass the reference gets from torch.autograd under
`loss.backward()` (apis/mmdet_train.py:72-79 through mmcv's OptimizerHook), issued here as explicit HIP backward kernels.

The forward mirror (lss.py, ...) calls `ops.*`; while a tape is active (`with Tape() as tape:`) those calls append a backward
closure.  `tape.backward()` runs the closures in reverse.  Gradients of activations live in buffers that mirror the
forward's activation STORAGE (one zero-initialised f32 buffer per storage, so the channel-offset "concat" writes and the
`t[:n]` views of the forward address the same gradient memory the same way); every backward op ACCUMULATES into its inputs'
gradient windows.  Parameter gradients are collected under the reference's state_dict names and layouts
(`tape.param_grads["img_encoder.img_backbone.layer1.0.conv1.weight"]` is [Cout, Cin, KH, KW]), which is what the gradient
golden F13 (tests/golden/gen_golden.py) is keyed by.

Every op of `EncoderDecoder.forward_train` has a recorder (both encoders, the fusion neck, the decoder with its look module
and teacher-forcing pass, and -- in losses.py -- the loss terms).  The recorders raise NotImplementedError for operand forms
they do not differentiate (row-run stem, unregistered weights, ...); the fused inference-only kernels (decoder_fused.py, the
*_ln look kernels) and the NCHW output conversions have no recorder and are not used by the taped forward.  trainer.Trainer drives the tape; tests/test_backward.py checks every sub-network against torch
autograd through the oracle and tests/test_train_step.py the whole model against the reference's own gradients (golden F13).
"""
import os

import torch

from . import ops

TAPE = None


class ConvMeta:
    """What a prepared Conv needs for its parameter gradients: state_dict names and the BatchNorm statistics folded into
    its scale / shift (scale = gamma / sigma, shift = beta - mean * scale [+ bias * scale]).  `kind` says how the prepared
    weight [Cout][KH][KW][cin_p] maps back to the reference parameter: "conv" [Cout, Cin, KH, KW]; "linear" [Cout, Cin];
    "cin_slice" = input channels [c0, c1) of a wider conv weight (`full_shape`); "dcn_group" = output rows [r0, r1) of the
    grouped deformable-conv weight, prepared as [og][1][9 taps][cg]."""

    def __init__(self, name, cin, bn=None, mean=None, sigma=None, bias=None, kind="conv", full_shape=None, lo=0):
        self.name, self.cin, self.bn, self.mean, self.sigma, self.bias = name, cin, bn, mean, sigma, bias
        self.kind, self.full_shape, self.lo = kind, full_shape, lo
        # the folded scale has (near-)zero entries (BatchNorm gamma ~ 0: zero_init_residual, pruned channels): dscale must
        # come from a recomputed raw convolution, not from dividing the saved output by the scale (refresh_small_scale_flags)
        self.scale_ref, self.small_scale = None, False

    def place_weight_grad(self, tape, dw):
        g = dw[..., :self.cin]
        if self.kind == "conv":
            tape.add_param_grad(self.name + ".weight", g.permute(0, 3, 1, 2).contiguous())
        elif self.kind == "linear":
            tape.add_param_grad(self.name + ".weight", g[:, 0, 0, :].contiguous())
        elif self.kind == "linear_hwc":    # Linear over a flattened (C, H*W) map whose columns were permuted to (H*W, C)
            o = g.shape[0]
            tape.add_param_grad(self.name + ".weight",
                                g[:, 0, 0, :].reshape(o, self.lo, -1).permute(0, 2, 1).reshape(o, -1).contiguous())
        elif self.kind == "cin_slice":
            full = tape.param_grads.get(self.name + ".weight")
            if full is None:
                full = torch.zeros(self.full_shape, dtype=torch.float32, device=dw.device)
                tape.param_grads[self.name + ".weight"] = full
            full[:, self.lo:self.lo + self.cin] += g.permute(0, 3, 1, 2)
        elif self.kind == "spconv":        # prepared [Cout][1][kD*kH*kW][Cin_p] -> spconv's (Cout, kD, kH, kW, Cin)
            tape.add_param_grad(self.name + ".weight", g[:, 0].reshape(self.full_shape).contiguous())
        elif self.kind == "dcn_group":
            full = tape.param_grads.get(self.name + ".weight")
            if full is None:
                full = torch.zeros(self.full_shape, dtype=torch.float32, device=dw.device)
                tape.param_grads[self.name + ".weight"] = full
            og = dw.shape[0]
            full[self.lo:self.lo + og] += g[:, 0].reshape(og, 3, 3, self.cin).permute(0, 3, 1, 2)
        else:
            raise NotImplementedError(self.kind)


class MetaTable:
    """tensor -> metadata, scoped to the model that registered it.

    Keyed by the tensor OBJECT: an entry holds a weak reference, is dropped when its tensor dies, and a lookup checks that the
    object found under an id is still the one that was registered -- a recycled id() can no longer hand a stale entry to a new
    tensor.  Every entry remembers its OWNER (the model whose load_state_dict was running: `owned_by`), so one model
    re-preparing its operands (`clear_metas(owner)`, trainer.Trainer._prepare) leaves every other model of the process alone
    (these were four process-global `id(tensor)` dicts)."""

    def __init__(self):
        self._d = {}

    def __setitem__(self, tensor, value):
        import weakref
        key = id(tensor)
        d = self._d
        self._d[key] = (weakref.ref(tensor, lambda _r, key=key, d=d: d.pop(key, None) if (d.get(key) or (None,))[0] is _r else None),
                        value, _OWNER[-1])

    def _entry(self, tensor):
        e = self._d.get(id(tensor))
        return e if e is not None and e[0]() is tensor else None

    def get(self, tensor, default=None):
        e = self._entry(tensor)
        return default if e is None else e[1]

    def __getitem__(self, tensor):
        e = self._entry(tensor)
        if e is None:
            raise KeyError("tensor is not registered")
        return e[1]

    def __contains__(self, tensor):
        return self._entry(tensor) is not None

    def __len__(self):
        return len(self._d)

    def items(self, owner=None):
        """(tensor, value) of the live entries (of `owner` only, if given)."""
        oid = None if owner is None else id(owner)
        out = []
        for ref, value, own in list(self._d.values()):
            t = ref()
            if t is not None and (oid is None or own == oid):
                out.append((t, value))
        return out

    def values(self, owner=None):
        return [v for _, v in self.items(owner)]

    def clear(self, owner=None):
        if owner is None:
            self._d.clear()
        else:
            oid = id(owner)
            for k in [k for k, e in self._d.items() if e[2] == oid]:
                del self._d[k]


_OWNER = [None]            # stack of id(model) whose operands are being prepared (owned_by)


class owned_by:
    """`with autodiff.owned_by(model):` -- registrations inside belong to `model` (EncoderDecoder.load_state_dict)."""

    def __init__(self, owner):
        self.oid = None if owner is None else id(owner)

    def __enter__(self):
        _OWNER.append(self.oid)

    def __exit__(self, *exc):
        _OWNER.pop()
        return False


AFFINE_META = MetaTable()   # scale tensor -> (bn name, mean, sigma), filled by layers.bn_affine
LN_META = MetaTable()       # gamma tensor -> (weight name, bias name) of a LayerNorm
CONV_META = MetaTable()     # weight tensor -> ConvMeta, filled by layers.conv_from_sd
PARAM_TENSORS = MetaTable() # tensor -> state_dict name: parameters the ops read directly as activations-like inputs
                            # (embeddings); their gradient buffers are moved to param_grads when backward() ends


def refresh_small_scale_flags(threshold=1e-4, owner=None):
    """One pass over the registered layers' folded scales (ONE host sync): flag those with |scale| < threshold * max|scale| in
    some channel.  trainer.Trainer calls it after (re)preparing the operands in frozen-BN mode."""
    metas = [m for m in CONV_META.values(owner) if m.scale_ref is not None]
    if not metas:
        return 0
    ratios = torch.stack([m.scale_ref.abs().min() / m.scale_ref.abs().max().clamp_min(1e-30) for m in metas]).cpu()
    for m, r in zip(metas, ratios.tolist()):
        m.small_scale = not (r >= threshold)          # (NaN compares false -> flagged)
    return sum(m.small_scale for m in metas)


def clear_metas(owner=None):
    """Forget the registered parameter mappings of `owner` (every model's when None) -- before that model's operands are
    re-prepared from new master weights."""
    for d in (CONV_META, AFFINE_META, LN_META, PARAM_TENSORS):
        d.clear(owner)


class paused:
    """`with autodiff.paused():` -- ops inside are not recorded (input plumbing, detached sub-graphs)."""

    def __enter__(self):
        global TAPE
        self.saved, TAPE = TAPE, None

    def __exit__(self, *exc):
        global TAPE
        TAPE = self.saved
        return False


def register_param(t, name):
    PARAM_TENSORS[t] = name
    return t


class _Keep(list):
    """The forward tensors the backward closures read, in recording order.  Remembers, per storage, the index of the FIRST node
    that referenced it (its producer, or for an input its first reader): once that node's backward has run no remaining node
    touches the storage, so Tape.backward releases the tensor and its gradient buffer there instead of at the end of the sweep
    (batch 8 at the thinktwice.py size: every activation AND every gradient buffer alive at once was 189 GB)."""

    def __init__(self, tape):
        super().__init__()
        self.tape = tape
        self.first = {}             # storage data_ptr -> node index

    def _reg(self, t):
        if isinstance(t, torch.Tensor):
            self.first.setdefault(t.untyped_storage().data_ptr(), len(self.tape.nodes))

    def append(self, t):
        self._reg(t)
        super().append(t)

    def __iadd__(self, items):
        items = list(items)
        for t in items:
            self._reg(t)
        return super().__iadd__(items)

    def clear(self):
        super().clear()
        self.first.clear()


class Tape:
    EAGER_RELEASE = os.environ.get("TT_TAPE_EAGER_RELEASE", "1") != "0"     # A/B knob (memory only; same arithmetic)

    def __init__(self, x3=True, release=False):
        """`release`: let backward() drop every activation and its gradient buffer as soon as the node that first referenced
        the storage has run (the training step; only parameter gradients survive the sweep).  Off by default: the sub-network
        tests read the gradients of graph inputs after backward()."""
        self.release = release and self.EAGER_RELEASE
        self.nodes = []
        self.grads = {}             # storage data_ptr -> flat f32 gradient buffer covering the whole storage
        self.param_grads = {}
        self.x3 = x3                # input gradients through the bf16x3 kernel (else exact f32)
        self._keep = _Keep(self)    # forward tensors the closures read: kept alive until their first node has run backward
        self._held = None           # release-mode sweep in progress: the per-storage table the references live in

    def __enter__(self):
        global TAPE
        assert TAPE is None, "nested tapes are not supported"
        TAPE = self
        return self

    def __exit__(self, *exc):
        global TAPE
        TAPE = None
        return False

    # ---- activation gradients
    def grad(self, t):
        """The gradient view of activation `t` (same shape / strides / storage offset inside its storage's buffer)."""
        assert t.dtype == torch.float32, "training runs on f32 activation storage (dtype torch.float32 or 'f32x3')"
        st = t.untyped_storage()
        key = st.data_ptr()
        buf = self.grads.get(key)
        if buf is None:
            buf = torch.zeros(st.nbytes() // 4, dtype=torch.float32, device=t.device)
            self.grads[key] = buf
            if self._held is not None:
                # created DURING a release-mode sweep (every activation whose gradient is not seeded by a loss): the reference
                # goes into the per-storage table, which drops it at the storage's first node -- appending to _keep here
                # would pin the activation until backward() returns
                self._held.setdefault(key, []).append(t)
            else:
                self._keep.append(t)
        return buf.as_strided(t.shape, t.stride(), t.storage_offset())

    def seed(self, t, g):
        """d(loss)/d(t) += g  (the loss side of the graph)."""
        self.grad(t).add_(g.to(t.device, torch.float32))

    def add_param_grad(self, name, g):
        if name in self.param_grads:
            self.param_grads[name] = self.param_grads[name] + g
        else:
            self.param_grads[name] = g

    def backward(self):
        global TAPE
        active, TAPE = TAPE, None          # the backward closures call forward ops too: nothing of that is recorded
        # storages to release after node i: those whose first reference is node i -- except parameter storages, whose gradient
        # buffers are collected below
        param_keys = {t.untyped_storage().data_ptr() for t, _ in PARAM_TENSORS.items()}
        release, held = {}, {}
        if self.release:
            for key, idx in self._keep.first.items():
                if key not in param_keys:
                    release.setdefault(idx, []).append(key)
            for t in self._keep:           # the references move into a per-storage table (the list is refilled by grad())
                if isinstance(t, torch.Tensor):
                    held.setdefault(t.untyped_storage().data_ptr(), []).append(t)
            list.clear(self._keep)
            self._held = held
        try:
            for i in range(len(self.nodes) - 1, -1, -1):
                self.nodes[i]()
                self.nodes[i] = None       # the closure holds forward tensors
                for key in release.get(i, ()):
                    self.grads.pop(key, None)
                    held.pop(key, None)
        finally:
            TAPE = active
            self._held = None
            held.clear()
        for t, name in PARAM_TENSORS.items():
            if t.is_cuda and t.untyped_storage().data_ptr() in self.grads:
                self.add_param_grad(name, self.grad(t).clone())
        self.nodes.clear()
        self._keep.clear()

    # ---- recorders (called from ops.* while the tape is active)
    def conv(self, x, w, y, stride, pad, dil, scale, shift, act, in_coff, cin, out_coff, res1, res1_coff, res2, res2_coff,
             pixel_shuffle2, in_cstride, shift_n=None, shift_n_mod=1, stop_grad=False, bn_raw=False):
        """`bn_raw`: the train-mode form of a BatchNorm'd layer -- this launch wrote the raw convolution (+ bias) and
        ops.batchnorm_train, recorded on its own, owns the BatchNorm parameters."""
        meta = CONV_META.get(w)
        if meta is None or in_cstride is not None:
            raise NotImplementedError("tape: convolution form without a backward yet (unnamed weight or row-run stem)")
        if pixel_shuffle2:
            return self._deconv2x2(meta, x, w, y, shift, act, in_coff, cin, out_coff, scale, res1, res2, bn_raw)
        Cout, KH, KW, cin_p = w.shape
        self._keep += [x, y, res1, res2, shift_n]
        inplace1 = res1 is not None and res1.data_ptr() == y.data_ptr() and res1_coff == out_coff
        if shift_n is not None and (res1 is not None or res2 is not None):
            raise NotImplementedError("tape: per-image shift together with residual inputs")

        def bwd():
            # (views with an image stride -- e.g. one time step of a (B, T, H, W, C) buffer -- are read through dense copies)
            gy, yd, xd = self.grad(y).contiguous(), y.contiguous(), x.contiguous()
            M = y.numel() // y.shape[-1]
            N, H, W_, _ = x.shape
            OH, OW = y.shape[1:3]
            g1 = None if res1 is None else self.grad(res1)
            g2 = None if res2 is None else self.grad(res2)
            if inplace1:
                # y overwrote res1: its old value is gone (so is dscale; such layers carry a bias only) and the gradient
                # w.r.t. it is g itself, written over gy in place
                assert scale is None and res2 is None, "in-place residual output: bias-only epilogue"
                dconv, _, dscale, dshift = ops.conv_epilogue_bwd(gy, yd, None, shift, act, None, None, C=Cout,
                                                                 dy_coff=out_coff, y_coff=out_coff, dres1=gy,
                                                                 dres1_coff=out_coff, dres_accumulate=False)
            elif shift_n is not None:
                # y = act(scale * conv + shift + shift_n[image]): the kernel's dscale = sum g * (pre - shift) / scale already
                # holds the shift_n term; d(shift_n)[image] = sum over the image's pixels of g
                gdense = torch.empty(M, Cout, dtype=torch.float32, device=y.device)
                dconv, _, dscale, dshift = ops.conv_epilogue_bwd(gy, yd, scale, shift, act, None, None, C=Cout,
                                                                 dy_coff=out_coff, y_coff=out_coff, dres1=gdense,
                                                                 dres_accumulate=False)
                imgs = shift_n.shape[0]
                assert shift_n_mod == imgs and N % imgs == 0, "tape: image n reads shift row n % rows"
                per_image = ops.spatial_pool(gdense.view(N, OH, OW, Cout), 0)
                self.grad(shift_n).add_(per_image.view(N // imgs, imgs, Cout).sum(0), alpha=float(OH * OW))
            else:
                pre = zraw = None
                if scale is not None and (meta.small_scale or act == 2):
                    # BatchNorm gamma ~ 0 in some channel / a sigmoid epilogue: dscale from the raw convolution itself
                    zraw = ops.conv2d(xd, w, stride=stride, pad=pad, dil=dil, act=0, in_coff=in_coff, cin=cin,
                                      out_dtype=torch.float32, _no_tape=True)
                if act not in (0, 1, 2):        # GELU / softplus: the derivative needs the pre-activation -> run the layer
                    pre = ops.conv2d(xd, w, stride=stride, pad=pad, dil=dil, scale=scale, shift=shift, act=0,   # once more
                                     in_coff=in_coff, cin=cin, res1=res1, res1_coff=res1_coff, res2=res2,
                                     res2_coff=res2_coff, _no_tape=True)
                dconv, _, dscale, dshift = ops.conv_epilogue_bwd(gy, yd, scale, shift, act, res1, res2, C=Cout,
                                                                 dy_coff=out_coff, y_coff=out_coff, res1_coff=res1_coff,
                                                                 res2_coff=res2_coff, dres1=g1, dres1_coff=res1_coff,
                                                                 dres2=g2, dres2_coff=res2_coff, pre=pre, conv_raw=zraw)
            dconv = dconv.view(N, OH, OW, Cout)
            # parameters
            dw = ops.conv2d_wgrad(xd, dconv, KH, KW, stride, pad, dil, cin=cin, in_coff=in_coff, cin_pad=cin_p, x3=self.x3)
            meta.place_weight_grad(self, dw)
            if bn_raw:
                if meta.bias is not None and shift is not None:
                    self.add_param_grad(meta.name + ".bias", dshift)
            elif meta.bn is not None:
                # scale = gamma / sigma, shift = beta + (bias - mean) * gamma / sigma
                off = meta.mean if meta.bias is None else meta.mean - meta.bias
                self.add_param_grad(meta.bn + ".weight", (dscale - off * dshift) / meta.sigma)
                self.add_param_grad(meta.bn + ".bias", dshift)
                if meta.bias is not None:
                    self.add_param_grad(meta.name + ".bias", dshift * scale)
            elif meta.bias is not None:
                self.add_param_grad(meta.name + ".bias", dshift)
            # input: conv of dconv with the rotated weights, accumulated into x's gradient window
            if stop_grad:           # the reference detaches this input (lss.py:589 seg_output.detach())
                return
            gx = self.grad(x)
            if KH == KW and gx.is_contiguous():
                ops.conv2d_dgrad(dconv, w, (H, W_), stride, pad, dil, x3=self.x3, out=gx, out_coff=in_coff)
            elif KH == KW:
                gx[..., in_coff:in_coff + cin] += ops.conv2d_dgrad(dconv, w, (H, W_), stride, pad, dil, x3=self.x3)
            else:
                # 1 x KW kernel over a 1 x KW "image" (the grouped deformable-conv GEMM over im2col columns): one 1x1
                # input-gradient GEMM per tap, written to that tap's pixel of the column gradient
                assert KH == 1 and H == 1 and W_ == KW and pad == 0 and stride == 1 and OH == 1 and OW == 1
                for t in range(KW):
                    d = ops.conv2d_dgrad(dconv, w[:, :, t:t + 1, :].contiguous(), (1, 1), 1, 0, 1, x3=False)
                    gx[:, 0, t, in_coff:in_coff + cin] += d.view(N, cin)

        self.nodes.append(bwd)

    def _deconv2x2(self, meta, x, w, y, shift, act, in_coff, cin, out_coff, scale, res1, res2, bn_raw=False):
        """ConvTranspose2d(k=2, s=2) = 1x1 GEMM to 4*Cout channels + pixel shuffle (layers.deconv2x2_from_sd): the
        gradient is un-shuffled (a layout copy) and the layer is differentiated as the 1x1 convolution it is."""
        if res1 is not None or res2 is not None:
            raise NotImplementedError("tape: transposed convolution with a residual input")
        C4 = w.shape[0]
        Cout = C4 // 4
        self._keep += [x, y]

        def bwd():
            N, H, W_, _ = x.shape
            # the epilogue (per real output channel: folded BN affine or bias, activation) in the shuffled layout
            gs, _, dscale, dshift = ops.conv_epilogue_bwd(self.grad(y), y, scale, shift, act, C=Cout, dy_coff=out_coff,
                                                          y_coff=out_coff)
            dconv = gs.view(N, H, 2, W_, 2, Cout).permute(0, 1, 3, 2, 4, 5).reshape(N, H, W_, C4).contiguous()
            dw = ops.conv2d_wgrad(x, dconv, 1, 1, 1, 0, 1, cin=cin, in_coff=in_coff, cin_pad=w.shape[-1], x3=self.x3)
            # prepared rows are (dh*2+dw)*Cout + co; the reference weight is [Cin, Cout, 2, 2]
            self.add_param_grad(meta.name + ".weight",
                                dw[:, 0, 0, :meta.cin].reshape(2, 2, Cout, meta.cin).permute(3, 2, 0, 1).contiguous())
            if bn_raw:
                if meta.bias is not None and shift is not None:
                    self.add_param_grad(meta.name + ".bias", dshift)
            elif meta.bn is not None:
                self.add_param_grad(meta.bn + ".weight", (dscale - meta.mean * dshift) / meta.sigma)
                self.add_param_grad(meta.bn + ".bias", dshift)
            elif meta.bias is not None:
                self.add_param_grad(meta.name + ".bias", dshift)
            ops.conv2d_dgrad(dconv, w, (H, W_), 1, 0, 1, x3=self.x3, out=self.grad(x), out_coff=in_coff)

        self.nodes.append(bwd)

    def layernorm_rows(self, x, gamma, beta, out, D, eps):
        names = LN_META.get(gamma)
        if names is None:
            raise NotImplementedError("tape: LayerNorm with unregistered parameters")
        self._keep += [x, out]

        def bwd():
            dg = torch.zeros(D, dtype=torch.float32, device=x.device)
            db = torch.zeros(D, dtype=torch.float32, device=x.device)
            ops.layernorm_rows_bwd(x, gamma, self.grad(out), self.grad(x), dg, db, D, eps)
            self.add_param_grad(names[0], dg)
            self.add_param_grad(names[1], db)

        self.nodes.append(bwd)

    # ---- look module (decoder): the waypoint / control inputs are detached by the reference (DEC:429-430)
    def look_gather_query(self, qos, ref, out, temporal, static, meas, flat, maps, row_stride):
        self._keep += [qos, ref, out, temporal, static, meas, flat] + list(maps)

        def bwd():
            B = meas.shape[0]
            grads = [self.grad(m) for m in maps]
            assert all(g.is_contiguous() for g in grads) and meas.is_contiguous() and flat.is_contiguous()
            ops.look_gather_query_bwd(B, qos, ref, self.grad(out), row_stride, self.grad(temporal), self.grad(static),
                                      self.grad(meas), self.grad(flat), grads)

        self.nodes.append(bwd)

    def msda_sample(self, value, offsets, logits, ref, level_hw, B, coff, out):
        self._keep += [value, offsets, logits, ref, out]

        def bwd():
            gv, go, gl = self.grad(value), self.grad(offsets), self.grad(logits)
            assert gv.is_contiguous() and go.is_contiguous() and gl.is_contiguous()
            ops.msda_sample_bwd(B, value, coff, offsets, logits, ref, level_hw, self.grad(out), gv, go, gl)

        self.nodes.append(bwd)

    def sca_reduce(self, x, max_len, B, out):
        self._keep += [x, max_len, out]

        def bwd():
            gx = self.grad(x)
            assert gx.is_contiguous()
            ops.sca_reduce_bwd(B, self.grad(out), max_len, gx)

        self.nodes.append(bwd)

    def value_shift(self, vshift, weight, weight_name, cams, cams_name, lvls, lvls_name):
        """vshift[l] (cam, 256) = (cams + lvls[l]) @ W^T -- the embedding part of value_proj(feat + cam + level) (DEC:392-393).
        Recorded BEFORE the projections that read it, so it runs after they have accumulated d(vshift)."""
        self._keep += list(vshift)

        def bwd():
            dW = torch.zeros_like(weight)
            dc = torch.zeros_like(cams)
            dl = torch.zeros_like(lvls)
            for l, vs in enumerate(vshift):
                g = self.grad(vs)                                   # (cam, 256 out)
                emb = cams.view(-1, 256) + lvls[l].view(1, 256)
                dW += g.t() @ emb
                de = g @ weight
                dc += de.view_as(dc)
                dl[l] += de.sum(0)
            self.add_param_grad(weight_name, dW)
            self.add_param_grad(cams_name, dc)
            self.add_param_grad(lvls_name, dl)

        self.nodes.append(bwd)

    def concat_rows(self, o2, pieces, coff):
        self._keep += [o2] + [p[0] for p in pieces]

        def bwd():
            go = self.grad(o2)
            c = coff
            for src, C, div, mod in pieces:
                if src is not None:
                    s2 = src.reshape(-1, src.shape[-1])
                    assert s2.data_ptr() == src.data_ptr()
                    ops.concat_piece_bwd(go, c, C, div, mod, self.grad(s2))
                c += C

        self.nodes.append(bwd)

    def copy_nhwc(self, x, out, C, in_coff, out_coff, rot_flip):
        self._keep += [x, out]

        def bwd():
            g = self.grad(out)[..., out_coff:out_coff + C]
            if rot_flip:        # out = rot90(flip(x, H), 1, (H, W))  (EDF:241,246)  =>  x-grad = flip(rot90(g, -1), H)
                g = torch.flip(torch.rot90(g, -1, (1, 2)), dims=[1])
            self.grad(x)[..., in_coff:in_coff + C] += g

        self.nodes.append(bwd)

    def ew(self, op, act, R, C, a2, a_coff, b2, b_coff, g2, g_coff, o2, out_coff):
        self._keep += [a2, b2, g2, o2]

        def bwd():
            gr = lambda t: None if t is None else self.grad(t)       # noqa: E731
            ops.ew_bwd(op, act, R, C, a2, a_coff, b2, b_coff, g2, g_coff, o2, out_coff, self.grad(o2), gr(a2), gr(b2),
                       gr(g2))

        self.nodes.append(bwd)

    def broadcast_rows(self, v, out, out_coff):
        self._keep += [v, out]
        self.nodes.append(lambda: ops.broadcast_rows_bwd(self.grad(out), self.grad(v), out_coff))

    def affine_rows(self, x, scale, shift, act, out):
        meta = AFFINE_META.get(scale)
        if meta is None and act == 0 and x.untyped_storage().data_ptr() not in self.grads:
            return      # a constant rescaling of a model input (speed / 12): nothing to differentiate
        if meta is None or act != 0:
            raise NotImplementedError("tape: affine_rows without a registered BatchNorm / with an activation")
        name, mean, sigma = meta
        self._keep += [x, out]

        def bwd():
            R, C = x.shape
            go = self.grad(out)
            # y = scale * x + shift on the first C columns: the conv-epilogue backward with conv := x
            dconv, _, dscale, dshift = ops.conv_epilogue_bwd(go, out, scale, shift, 0, C=C)
            self.add_param_grad(name + ".weight", (dscale - mean * dshift) / sigma)
            self.add_param_grad(name + ".bias", dshift)
            self.grad(x).add_(dconv)

        self.nodes.append(bwd)

    def channel_gate(self, x, gate, res, out, gate_act, out_act):
        from . import _lib
        plain = res is None and out_act == _lib.ACT_NONE
        se_block = res is not None and out_act == _lib.ACT_RELU      # SEBasicBlock: relu(x * sigmoid(g) + shortcut)
        if gate_act != _lib.ACT_SIGMOID or not (plain or se_block):
            raise NotImplementedError("tape: this channel-gate form has no backward yet")
        self._keep += [x, gate, out, res]
        if plain:
            self.nodes.append(lambda: ops.channel_gate_bwd(x, gate, self.grad(out).contiguous(), self.grad(x),
                                                           self.grad(gate)))
        else:
            self.nodes.append(lambda: ops.channel_gate_bwd(x, gate, self.grad(out).contiguous(), self.grad(x),
                                                           self.grad(gate), out_relu=out, dres=self.grad(res)))

    def spatial_pool(self, x, out, mode, C, coff):
        self._keep += [x, out]
        if mode == 0:
            self.nodes.append(lambda: ops.spatial_mean_bwd(self.grad(out).contiguous(), self.grad(x), C, coff))
        else:
            assert coff == 0 and C == x.shape[-1], "tape: (mean + amax) / 2 pooling over a channel window"
            self.nodes.append(lambda: ops.spatial_meanmax_bwd(x, self.grad(out).contiguous(), self.grad(x)))

    def repeat_rows(self, t, out, times):
        self._keep += [t, out]
        self.nodes.append(lambda: self.grad(t).add_(self.grad(out).view(times, *t.shape).sum(0)))

    def deform_im2col3x3(self, x, offsets, cols, pad):
        self._keep += [x, offsets, cols]
        self.nodes.append(lambda: ops.deform_im2col3x3_bwd(x, offsets, self.grad(cols).contiguous(), self.grad(x),
                                                           self.grad(offsets), pad))

    def gather_conv(self, feats, nbr, m_dev, w, scale, shift, act, res, out, in_rows, bn_raw=False):
        """Sparse convolution (rulebook GEMM) + BatchNorm1d + residual + ReLU.  `in_rows`: None for a submanifold layer
        (input rows == output rows), else (device row count, allocated rows) of the INPUT level of a strided layer."""
        meta = CONV_META.get(w)
        if meta is None:
            raise NotImplementedError("tape: sparse convolution with an unregistered weight")
        Cout, _, taps, cin_p = w.shape
        self._keep += [feats, nbr, m_dev, res, out]

        def bwd():
            gy = self.grad(out)
            g1 = None if res is None else self.grad(res)
            dconv, _, dscale, dshift = ops.conv_epilogue_bwd(gy, out, scale, shift, act, res, None, C=Cout, dres1=g1,
                                                             m_dev=m_dev)
            dw = ops.gather_conv_wgrad(feats, nbr, m_dev, dconv, taps, cin_pad=cin_p)
            meta.place_weight_grad(self, dw)
            if not bn_raw:
                self.add_param_grad(meta.bn + ".weight", (dscale - meta.mean * dshift) / meta.sigma)
                self.add_param_grad(meta.bn + ".bias", dshift)
            # input rows: the same gathered GEMM on the transposed rulebook, accumulated into the input's gradient -- in bf16x3
            # like the dense input gradients when the tape runs in that mode (a submanifold layer's transposed rulebook is a
            # submanifold rulebook again: the run-staged kernel takes it; was the exact-f32 register-staged gather kernel,
            # 47 ms per iteration at batch 8)
            from . import weights
            gx = self.grad(feats)
            if in_rows is None:     # submanifold: nbr[m][t] = j  <=>  nbr[j][taps - 1 - t] = m
                wt = w.flip(2).permute(3, 1, 2, 0).contiguous()
                ops.gather_conv(dconv, nbr, m_dev, wt, res=gx, out=gx, _no_tape=True,
                                w_x3=weights.split_pairs_x3(wt) if self.x3 else None)
            else:
                rows_dev, rows_max = in_rows
                inv = ops.sp_inverse_rulebook(nbr, m_dev, rows_max)
                wt = w.permute(3, 1, 2, 0).contiguous()
                ops.gather_conv(dconv, inv, rows_dev, wt, res=gx, out=gx, _no_tape=True, stride=2,
                                w_x3=weights.split_pairs_x3(wt) if self.x3 else None)

        self.nodes.append(bwd)

    def bn_train(self, z, out, out_coff, spec, act, res1, res1_coff, res2, res2_coff, m_dev, groups, stats, scale, mean, invstd):
        """Train-mode BatchNorm (ops.batchnorm_train): batch-statistics backward with its two reduction terms; dgamma / dbeta
        under the layer's state_dict names; under SyncBN the reductions are all-reduced inside ops.batchnorm_train_bwd."""
        self._keep += [z, out, res1, res2, stats, scale, mean, invstd, m_dev]

        def bwd():
            gy = self.grad(out)
            assert gy.is_contiguous(), "bn_train: the output's gradient buffer must be row-linear"
            g1 = None if res1 is None else self.grad(res1)
            g2 = None if res2 is None else self.grad(res2)
            for g in (g1, g2):
                assert g is None or g.is_contiguous()
            dz = self.grad(z)
            dgamma, dbeta = ops.batchnorm_train_bwd(gy, out_coff, out, out_coff, z, mean, invstd, scale, stats, act, g1,
                                                    res1_coff, g2, res2_coff, dz, m_dev=m_dev, groups=groups)
            if spec.gamma is not None:
                self.add_param_grad(spec.name + ".weight", dgamma)
                self.add_param_grad(spec.name + ".bias", dbeta)

        self.nodes.append(bwd)

    def dropout(self, x, out, mask, p):
        self._keep += [x, out, mask]
        self.nodes.append(lambda: ops.dropout_bwd(self.grad(out), mask, self.grad(x), p))

    def sp_to_dense(self, x, coords, rows, max_rows, dims, dense):
        self._keep += [x, coords, rows, dense]
        self.nodes.append(lambda: ops.sp_from_dense(self.grad(dense), coords, rows, max_rows, dims, self.grad(x)))

    def lift_splat(self, depth_logits, context, geom, voxel_num, B, ncam, out, out_coff, rot_flip):
        if rot_flip:
            raise NotImplementedError("tape: lift-splat with the fused rot90/flip output")
        self._keep += [depth_logits, context, geom, out]
        self.nodes.append(lambda: ops.lift_splat_bwd(depth_logits, context, geom, voxel_num, B, ncam, self.grad(out),
                                                     out_coff, self.grad(depth_logits), self.grad(context)))

    def bilinear_up2(self, x, y):
        self._keep += [x, y]
        self.nodes.append(lambda: ops.bilinear_up2_bwd(self.grad(y).contiguous(), self.grad(x)))

    def maxpool3x3s2(self, x, y):
        self._keep += [x, y]
        self.nodes.append(lambda: ops.maxpool3x3s2_bwd(x, self.grad(y).contiguous(), self.grad(x)))

    def upsample_nearest_add_(self, dst, src):
        self._keep += [dst, src]
        # dst += up(src) in place: d/d(old dst) is the identity (the gradient buffer is shared), d/d(src) pools it
        self.nodes.append(lambda: ops.upsample_nearest_add_bwd(self.grad(dst).contiguous(), self.grad(src)))
