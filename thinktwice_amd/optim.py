"""AdamW + global-norm gradient clipping over the flat buffers of `grad_sync.FlatGradBuffer` (the reference's
`optimizer` / `optimizer_config`, configs/thinktwice.py:282-287), two HIP launches per step, no host sync."""
import ctypes

import torch

from ._lib import check, cur_stream, lib, ptr, require_cuda

_f = ctypes.c_float


class FlatAdamW:
    def __init__(self, flat_param, flat_grad, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7,
                 max_grad_norm=100.0):
        require_cuda(flat_param, flat_grad)
        assert flat_param.dtype == flat_grad.dtype == torch.float32 and flat_param.is_contiguous()
        assert flat_param.numel() == flat_grad.numel() and flat_grad.is_contiguous()
        self.p, self.g = flat_param, flat_grad
        self.m = torch.zeros_like(flat_param)
        self.v = torch.zeros_like(flat_param)
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.steps = 0
        self._ws = torch.empty(1024, dtype=torch.float32, device=flat_param.device)
        self.norm_scale = torch.zeros(2, dtype=torch.float32, device=flat_param.device)   # [||g||, clip factor]

    def step(self, lr=None):
        """One optimizer step; returns the (device) tensor [grad norm, clip factor] of this step."""
        n = ctypes.c_longlong(self.p.numel())
        st = cur_stream(self.p.device)
        scale = None
        if self.max_norm is not None:
            check(lib().tt_grad_norm_clip(ptr(self.g), n, _f(self.max_norm), ptr(self._ws), ptr(self.norm_scale), st),
                  "tt_grad_norm_clip")
            scale = ctypes.c_void_p(self.norm_scale.data_ptr() + 4)
        self.steps += 1
        check(lib().tt_adamw_step(ptr(self.p), ptr(self.g), ptr(self.m), ptr(self.v), n,
                                  _f(self.lr if lr is None else lr), _f(self.betas[0]), _f(self.betas[1]),
                                  _f(self.eps), _f(self.wd), ctypes.c_int(self.steps),
                                  scale if scale is not None else ctypes.c_void_p(0), st), "tt_adamw_step")
        return self.norm_scale
