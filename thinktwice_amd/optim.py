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
        # the count of APPLIED updates lives on the device (tt_adamw_advance): an iteration whose gradient norm is not finite
        # is skipped there, and the bias corrections / the checkpointed `step` must stay with the untouched moments
        self.steps_dev = torch.zeros(1, dtype=torch.int32, device=flat_param.device)
        self.issued, self._base = 0, 0   # optimizer steps issued by the host since the count was set (>= the applied ones)
        self._ws = torch.empty(1024, dtype=torch.float32, device=flat_param.device)
        self.norm_scale = torch.zeros(2, dtype=torch.float32, device=flat_param.device)   # [||g||, clip factor]

    @property
    def steps(self):
        """Updates actually applied (one blocking 4-byte read: checkpoints and tests, not the training loop)."""
        return int(self.steps_dev.item())

    @steps.setter
    def steps(self, n):
        self.steps_dev.fill_(int(n))
        self._base, self.issued = int(n), 0

    def skipped(self):
        """Issued optimizer steps the device did NOT apply (non-finite gradient norm) since the count was last set.  Blocking."""
        return self.issued - (self.steps - self._base)

    def state_dict(self):
        """Adam moments + step count (what an mmcv checkpoint's `optimizer` entry carries for a resume)."""
        return {"exp_avg": self.m.detach().clone().cpu(), "exp_avg_sq": self.v.detach().clone().cpu(), "step": self.steps,
                "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd}

    def load_state_dict(self, sd):
        self.m.copy_(sd["exp_avg"].to(self.m.device))
        self.v.copy_(sd["exp_avg_sq"].to(self.v.device))
        self.steps = int(sd["step"])

    def step(self, lr=None, live_ranges=None):
        """One optimizer step; returns the (device) tensor [grad norm, clip factor] of this step.  `live_ranges`: [(offset,
        count)] element ranges to update -- torch's AdamW skips parameters whose .grad is None (the reference's 90 dead
        ones: no weight decay, no moment update); the ranges of the others, merged, are what this step touches."""
        n = ctypes.c_longlong(self.p.numel())
        st = cur_stream(self.p.device)
        scale = None
        if self.max_norm is not None:
            check(lib().tt_grad_norm_clip(ptr(self.g), n, _f(self.max_norm), ptr(self._ws), ptr(self.norm_scale), st),
                  "tt_grad_norm_clip")
            scale = ctypes.c_void_p(self.norm_scale.data_ptr() + 4)
        self.issued += 1
        sc = scale if scale is not None else ctypes.c_void_p(0)
        for off, cnt in (live_ranges if live_ranges is not None else [(0, self.p.numel())]):
            at = lambda t: ctypes.c_void_p(t.data_ptr() + 4 * off)
            check(lib().tt_adamw_step_dev(at(self.p), at(self.g), at(self.m), at(self.v), ctypes.c_longlong(cnt),
                                          _f(self.lr if lr is None else lr), _f(self.betas[0]), _f(self.betas[1]),
                                          _f(self.eps), _f(self.wd), ptr(self.steps_dev), sc, st), "tt_adamw_step_dev")
        check(lib().tt_adamw_advance(ptr(self.steps_dev), sc, st), "tt_adamw_advance")
        return self.norm_scale


def warmup_cosine_lr(base_lr, it, total_iters, warmup_iters=1000, warmup_ratio=1.0 / 3, min_lr_ratio=1e-3,
                     iters_per_epoch=None, by_epoch=True):
    """Learning rate of iteration `it` (0-based) under the reference's lr_config (configs/thinktwice.py:286-291: mmcv
    CosineAnnealingLrUpdaterHook with linear warmup, under `EpochBasedRunner`).  mmcv's hook anneals BY EPOCH by default
    (`by_epoch=True`): the regular rate is constant within an epoch, base_lr -> base_lr * min_lr_ratio along
    cos(pi * epoch / max_epochs); only the WARMUP runs per iteration (`warmup_by_epoch=False`): during the first `warmup_iters`
    iterations the regular rate is scaled by 1 - (1 - it / warmup_iters) * (1 - warmup_ratio).
    `iters_per_epoch` gives the epoch of an iteration (epoch = it // iters_per_epoch, max_epochs = ceil(total_iters /
    iters_per_epoch)); without it -- or with by_epoch=False -- the cosine is evaluated per iteration (mmcv's by_epoch=False)."""
    import math
    if by_epoch and not iters_per_epoch:
        raise ValueError("warmup_cosine_lr(by_epoch=True) -- the reference's schedule -- needs iters_per_epoch; pass "
                         "by_epoch=False for mmcv's per-iteration cosine")
    target = base_lr * min_lr_ratio
    if by_epoch:
        progress = min(it, total_iters) // iters_per_epoch
        max_progress = max(1, -(-total_iters // iters_per_epoch))
    else:
        progress, max_progress = min(it, total_iters), max(1, total_iters)
    lr = target + 0.5 * (base_lr - target) * (1.0 + math.cos(math.pi * progress / max_progress))
    if it < warmup_iters:
        lr *= 1.0 - (1.0 - it / warmup_iters) * (1.0 - warmup_ratio)
    return lr
