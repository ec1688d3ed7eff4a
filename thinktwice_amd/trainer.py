"""One data-parallel training iteration (reference: mmcv EpochBasedRunner.run_iter -> model.train_step
(encoder_decoder_framework.py:140-145) + OptimizerHook (loss.backward(), grad clip, AdamW.step; configs/thinktwice.py:282-290)
under MMDistributedDataParallel (apis/mmdet_train.py:67-74)).

    forward_train on the HIP forward  ->  reverse sweep of the tape (thinktwice_amd/autodiff.py: hand-written backward kernels,
    gradients under the reference's state_dict names)  ->  ONE all-reduce of the flat gradient buffer (grad_sync.py, RCCL)
    ->  global-norm clip + AdamW on the flat buffers (optim.py, two launches)  ->  operands re-prepared from the master weights.

The master parameters are f32 views of one flat buffer (288 GB of HBM: master weights, Adam moments, the flat gradient and the
kernels' operand formats all stay resident).  BatchNorm runs on its running statistics (frozen-BN fine-tuning, what golden F10 /
F13 pin); its affine parameters train, its statistics are not updated.  Parameters the loss never reaches (the reference's 90
dead ones) keep zero gradients, as under DDP's find_unused_parameters.
"""
import os
import sys

import torch

from . import autodiff
from .grad_sync import FlatGradBuffer
from .optim import FlatAdamW


# registered buffers of the reference modules (BatchNorm statistics; the LSS frustum / voxel grid constants, lss.py:470-476)
_BUFFERS = ("running_mean", "running_var", "num_batches_tracked", "voxel_size", "voxel_coord", "voxel_num", "frustum")


def _trainable(name, t):
    return torch.is_tensor(t) and t.is_floating_point() and t.dim() > 0 and not name.endswith(_BUFFERS)


class Trainer:
    def __init__(self, model, state_dict, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7, max_grad_norm=100.0,
                 x3=None, frozen_bn=True):
        """`model`: an EncoderDecoder (dtype torch.float32 or "f32x3"); `state_dict`: reference-format weights.  `x3`: input
        gradients of the convolutions through the bf16x3 kernel (default: when the model's forward uses it).
        `frozen_bn=False`: the reference's training semantics (model.train(): batch-statistics BatchNorm with SyncBN across
        the ranks, running statistics updated in `self.buffers`, live ASPP dropout; goldens F11 / F16).  `frozen_bn=True`:
        BatchNorm on its running statistics (model.eval() arithmetic, goldens F10 / F13) -- fine-tuning with frozen
        statistics; the BatchNorm affine parameters still train."""
        dev = model.device
        self.model = model
        self.frozen_bn = bool(frozen_bn)
        self.x3 = (model.dtype != torch.float32) if x3 is None else x3
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items() if k != "_metadata"}
        self.names = [k for k, v in sd.items() if _trainable(k, v)]
        self._trainable = set(self.names)
        total = sum(sd[k].numel() for k in self.names)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.sd = {}
        off = 0
        for k, v in sd.items():
            if k in self._trainable:
                view = self.flat_param[off:off + v.numel()].view(v.shape)
                view.copy_(v.to(dev, torch.float32))
                self.sd[k] = view.requires_grad_(True)          # a leaf whose .grad FlatGradBuffer points into its buffer
                off += v.numel()
            else:
                self.sd[k] = v
        self.grads = FlatGradBuffer([self.sd[k] for k in self.names])
        self.opt = FlatAdamW(self.flat_param, self.grads.flat, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                             max_grad_norm=max_grad_norm)
        self.param_grads = None
        self._stepped = set()          # names of the parameters the optimizer has updated at least once (torch: those with state)
        self.iteration, self.epoch, self._bn_steps = 0, 0, 0
        self._skips_accounted = 0      # device-skipped updates already taken out of iteration / _bn_steps (reconcile())
        self.schedule = None           # set_schedule(): the reference's lr_config
        self._prepare_on_device = os.environ.get("TT_TRAIN_PREPARE", "device") != "host"
        # BatchNorm running statistics: device copies the prepared layers alias (train mode updates them in place)
        self.buffers = {k: v.to(dev, torch.float32).contiguous() for k, v in self.sd.items()
                        if torch.is_tensor(v) and k.endswith(("running_mean", "running_var"))}
        self._dev_buffers = self.buffers
        if not self.frozen_bn and not self._prepare_on_device:
            raise ValueError("Trainer(frozen_bn=False) keeps the running statistics on the device: TT_TRAIN_PREPARE=host "
                             "is a frozen-BN path")
        self._prepare()

    def _prepare(self):
        """(Re)build the kernels' operand formats (folded BatchNorm affines, channel-last / pair-split weights) from the
        master weights.  Device path: the load code runs on views of the flat master buffer (BatchNorm statistics moved to
        the device once), so nothing crosses PCIe.  TT_TRAIN_PREPARE=host takes the route a checkpoint takes instead (one
        0.5 GB device-to-host copy + host-side preparation per iteration); it is also the fallback if the load code meets
        a host-only operation on a device tensor."""
        autodiff.clear_metas(self.model)
        with torch.no_grad():
            if self._prepare_on_device:
                try:
                    self.model.load_state_dict({k: (v.detach() if k in self._trainable else self._dev_buffers.get(k, v))
                                                for k, v in self.sd.items()})
                    if self.frozen_bn:
                        autodiff.refresh_small_scale_flags(owner=self.model)
                    return
                except (RuntimeError, TypeError) as e:
                    print(f"[trainer] device-side operand preparation failed ({type(e).__name__}: {e}); "
                          f"using the host path", file=sys.stderr, flush=True)
                    self._prepare_on_device = False
                    autodiff.clear_metas(self.model)
            self.model.load_state_dict({k: (v.detach().cpu() if k in self._trainable else v) for k, v in self.sd.items()})

    def backward(self, batch):
        """Forward + losses + reverse sweep: fills the flat gradient buffer (no collective, no update).  Returns
        dict(loss, log_vars, num_samples) of train_step."""
        self.grads.zero_()
        was_training = self.model.training
        self.model.train(not self.frozen_bn)
        try:
            with autodiff.Tape(x3=self.x3, release=True) as tape:
                out = self.model.train_step(batch, None)
                tape.backward()
        finally:
            self.model.train(was_training)
        self.param_grads = tape.param_grads
        # element ranges of the parameters that received a gradient (the others are skipped by the optimizer like torch's
        # grad-is-None parameters): merged runs in flat-buffer order
        runs, off = [], 0
        for k in self.names:
            n = self.sd[k].numel()
            if k in tape.param_grads:
                if runs and runs[-1][0] + runs[-1][1] == off:
                    runs[-1][1] += n
                else:
                    runs.append([off, n])
            off += n
        self.live_ranges = [tuple(r) for r in runs]
        unknown = [k for k in tape.param_grads if k not in self.sd]
        assert not unknown, f"gradients for names outside the state_dict: {unknown[:5]}"
        for k, g in tape.param_grads.items():
            self.sd[k].grad.copy_(g.reshape(self.sd[k].shape))
        return out

    def set_schedule(self, total_iters, iters_per_epoch, warmup_iters=1000, warmup_ratio=1.0 / 3, min_lr_ratio=1e-3):
        """The reference's `lr_config` (configs/thinktwice.py:286-291: cosine annealing BY EPOCH with a per-iteration linear
        warm-up, optim.warmup_cosine_lr): from now on step() without an explicit `lr` uses the rate of `self.iteration`, and
        `self.epoch` follows the iteration count (EpochBasedRunner)."""
        if not iters_per_epoch or iters_per_epoch < 1:
            raise ValueError("set_schedule: iters_per_epoch (len(data_loader)) is required for the by-epoch cosine")
        self.schedule = dict(total_iters=int(total_iters), iters_per_epoch=int(iters_per_epoch), warmup_iters=int(warmup_iters),
                             warmup_ratio=float(warmup_ratio), min_lr_ratio=float(min_lr_ratio))

    def current_lr(self):
        if self.schedule is None:
            return self.opt.lr
        from .optim import warmup_cosine_lr
        return warmup_cosine_lr(self.opt.lr, self.iteration, **self.schedule)

    def reconcile(self):
        """Blocking.  With check_finite=False the host runs ahead of the device, which skips the update of an iteration whose
        gradient norm is not finite (p, m, v and the applied-step count untouched; non-finite batch statistics never reach
        the BatchNorm running buffers either, tt_bn_finalize).  This takes those skipped iterations back out of the host's
        counters (`iteration`, the BatchNorm call count) so that checkpoints agree with the device; called by state_dict() /
        optimizer_state_dict() / checkpoint().  Returns the number of skipped updates found."""
        k = self.opt.skipped() - self._skips_accounted
        if k > 0:
            self.iteration -= k
            if not self.frozen_bn:
                self._bn_steps -= k
            self._skips_accounted += k
            if self.schedule is not None:
                self.epoch = self.iteration // self.schedule["iters_per_epoch"]
        return max(k, 0)

    def step(self, batch, lr=None, check_finite=True):
        """One iteration; adds `grad_norm` (device tensor [norm, clip factor]) to train_step's dict.  A non-finite gradient
        norm never reaches the weights: tt_grad_norm_clip hands the update a NaN factor and tt_adamw_step_dev skips (p, m, v
        and the device-side step count untouched).  `check_finite` (one host sync per iteration) additionally raises here;
        pass False to stay asynchronous and look at `grad_norm` whenever convenient (reconcile() squares the host counters
        with what the device applied).  Without `lr`, the rate of set_schedule() (else the constructor's)."""
        if lr is None and self.schedule is not None:
            lr = self.current_lr()
        out = self.backward(batch)
        self.grads.all_reduce_mean()
        out["grad_norm"] = self.opt.step(lr, live_ranges=self.live_ranges)
        if check_finite and not bool(torch.isfinite(out["grad_norm"][0])):
            self._skips_accounted += 1              # nothing below runs: iteration / BatchNorm counters never saw this step
            raise FloatingPointError("Trainer.step: non-finite gradient norm; the update was skipped on the device (weights, "
                                     "Adam moments and running statistics of the optimizer are those of the last good step)")
        self._stepped.update(self.param_grads.keys())
        self.iteration += 1
        if not self.frozen_bn:
            self._bn_steps += 1
        if self.schedule is not None:
            self.epoch = self.iteration // self.schedule["iters_per_epoch"]
        self._prepare()
        return out

    # ---- checkpoints (mmcv CheckpointHook / torch.save layout: meta + state_dict + optimizer)
    def _num_batches_tracked(self, key, value):
        """nn.BatchNorm's call counter under model.train(): every BatchNorm inside the per-sweep camera pass is called once
        per SWEEP and iteration (lss.py:689-714; older sweeps run under no_grad but in train mode), the others once per
        iteration."""
        if self.frozen_bn or self._bn_steps == 0:
            return value
        T = int((self.model.config or {}).get("queue_length", 1))
        per_iter = T if (key.startswith("img_encoder.") and "bev_multiframe_merge" not in key) else 1
        return value + self._bn_steps * per_iter

    def state_dict(self):
        """Reference-format weights (own copies: the master tensors are views of one flat buffer), BatchNorm running
        statistics and `num_batches_tracked` as they stand now."""
        self.reconcile()
        out = {}
        for k, v in self.sd.items():
            if k in self._trainable:
                out[k] = v.detach().clone()
            elif k in self.buffers:
                out[k] = self.buffers[k].detach().clone().cpu()
            elif k.endswith("num_batches_tracked") and torch.is_tensor(v):
                out[k] = self._num_batches_tracked(k, v.clone())
            else:
                out[k] = v
        return out

    def optimizer_state_dict(self):
        """torch.optim.AdamW.state_dict() layout (what mmcv saves under 'optimizer' and `optimizer.load_state_dict` takes on
        resume, train.py:238 / EpochBasedRunner.resume): per-parameter state keyed by the parameter's index in
        model.parameters() order (= state_dict order without the buffers), plus one param_group.  Parameters that never
        received a gradient carry no state, exactly like torch's `grad is None` ones (the reference's 90 dead parameters)."""
        self.reconcile()
        o = self.opt
        steps = o.steps
        state, off = {}, 0
        for idx, k in enumerate(self.names):
            n = self.sd[k].numel()
            if k in self._stepped:
                shape = self.sd[k].shape
                state[idx] = {"step": torch.tensor(float(steps)),
                              "exp_avg": o.m[off:off + n].view(shape).detach().clone().cpu(),
                              "exp_avg_sq": o.v[off:off + n].view(shape).detach().clone().cpu()}
            off += n
        # mmcv's LrUpdaterHook stores the BASE rate under 'initial_lr' and overwrites 'lr' with the scheduled (warmed-up /
        # annealed) one every iteration: write both, so a resume by either side takes the schedule's base from 'initial_lr'
        group = {"lr": self.current_lr(), "initial_lr": o.lr, "betas": tuple(o.betas), "eps": o.eps, "weight_decay": o.wd,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "params": list(range(len(self.names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, osd):
        o = self.opt
        self._skips_accounted = 0                  # (setting the step count restarts the issued / applied bookkeeping)
        if "exp_avg" in osd:                       # round-3 flat layout (whole-buffer moments)
            o.load_state_dict(osd)
            self._stepped = set(self.names)
            return
        o.m.zero_()
        o.v.zero_()
        self._stepped = set()
        off, steps = 0, 0
        for idx, k in enumerate(self.names):
            n = self.sd[k].numel()
            st = osd["state"].get(idx, osd["state"].get(str(idx)))
            if st is not None:
                o.m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(o.m.device, torch.float32))
                o.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(o.v.device, torch.float32))
                steps = max(steps, int(float(st["step"])))
                self._stepped.add(k)
            off += n
        o.steps = steps
        g = (osd.get("param_groups") or [{}])[0]
        # the schedule's BASE rate: mmcv checkpoints carry it as 'initial_lr' ('lr' there is the already scheduled rate of the
        # iteration the checkpoint was written at -- using it as the base would apply the schedule twice)
        o.lr = g.get("initial_lr", g.get("lr", o.lr))
        o.betas = tuple(g.get("betas", o.betas))
        o.eps, o.wd = g.get("eps", o.eps), g.get("weight_decay", o.wd)

    def checkpoint(self, epoch=None):
        """What an mmcv checkpoint holds for a resume (configs/thinktwice.py:292 checkpoint_config, mmcv save_checkpoint):
        `meta` (epoch / iter), `state_dict` (reference key names), `optimizer` (torch AdamW layout)."""
        self.reconcile()
        return {"meta": {"epoch": self.epoch if epoch is None else epoch, "iter": self.iteration},
                "state_dict": self.state_dict(), "optimizer": self.optimizer_state_dict()}

    def load_checkpoint(self, ckpt):
        sd = ckpt["state_dict"]
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        with torch.no_grad():
            for k in self.names:
                self.sd[k].copy_(sd[k].to(self.sd[k].device, torch.float32))
            for k, b in self.buffers.items():
                b.copy_(sd[k].to(b.device, torch.float32))
            for k, v in sd.items():
                if k.endswith("num_batches_tracked") and k in self.sd:
                    self.sd[k] = v.clone()
        self._bn_steps = 0
        meta = ckpt.get("meta") or {}
        self.epoch, self.iteration = int(meta.get("epoch", 0) or 0), int(meta.get("iter", 0) or 0)
        if "optimizer" in ckpt:
            self.load_optimizer_state_dict(ckpt["optimizer"])
        else:
            # no optimizer state: rebase the issued / applied step bookkeeping where it stands, so that updates skipped BEFORE
            # this load are not taken out of the freshly loaded iteration count by the next reconcile()
            self.opt.steps = self.opt.steps
            self._skips_accounted = 0
        self._prepare()
