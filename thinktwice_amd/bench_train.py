"""bench.py --workload train_step: one data-parallel training iteration (trainer.Trainer.step) at the thinktwice.py
configuration -- forward_train, the tape's reverse sweep, ONE all-reduce of the flat gradient buffer over RCCL (world > 1),
global-norm clip + AdamW, operand re-preparation.  model.train() semantics by default (batch-statistics BatchNorm with the
SyncBN statistic exchange, live ASPP dropout: the reference's training mode; TT_BENCH_TRAIN_FROZEN_BN=1 selects the
running-statistics fine-tuning mode), synthetic batch and targets, random-init weights.  Not the BASELINE.json metric (that
is the inference forward): BASELINE config 4's single-GPU number, also reported as the `train_step` leg of the default line."""
import os
import time

import torch

from . import model as tm
from . import ops, params, synth
from .bench_forward import MFMA_PEAK_TF, TORCH_DTYPE


class TrainStepWorkload:
    metric = "training samples/sec (forward_train + backward + all-reduce + clip/AdamW, thinktwice.py cfg)"

    def __init__(self, batch, device, dtype=None, frozen_bn=None):
        from .trainer import Trainer
        dtype = dtype or os.environ.get("TT_BENCH_DTYPE", "bf16x3")
        assert dtype in ("f32", "bf16x3"), "training runs on f32 activation storage"
        self.dtype, self.B = dtype, batch
        self.name = (f"train_step thinktwice.py cfg: {batch} samples x (2 sweeps x 4 cams x 448x896 + 65536-pt LiDAR), "
                     f"23 loss terms incl. the teacher-forcing pass, 878 live / 968 parameters")
        self.precision_note = ("f32 storage; forward, input-gradient and (>= 64 channels a side) weight-gradient convolutions in "
                               "bf16x3, the remaining weight gradients on the exact-f32 MFMA"
                               if dtype == "bf16x3" else "f32 everywhere (exact-f32 MFMA)")
        self.launch_note = "eager launches, one stream"
        self.model, self.cfg = tm.build_thinktwice(dtype=TORCH_DTYPE[dtype], device=str(device))
        sd = params.init_params(self.cfg, seed=0)
        if frozen_bn is None:
            frozen_bn = os.environ.get("TT_BENCH_TRAIN_FROZEN_BN", "0") == "1"
        self.frozen_bn = frozen_bn
        self.trainer = Trainer(self.model, sd, frozen_bn=frozen_bn)
        del sd
        rank = int(os.environ.get("RANK", "0"))
        self.batch = tm.batch_to_device(synth.make_batch(batch, seed=1234 + rank * batch), device)
        self.batch.update(synth.make_train_targets(batch))
        self.last = None

    def step(self):
        self.last = self.trainer.step(self.batch)
        return self.last

    def frames_per_step(self):
        return self.B

    def collect(self):
        """Runs on EVERY rank after the timed steps (the profiled iteration issues the step's collectives): all
        convolution-shaped launches of one iteration (forward, GELU recomputes, input gradients, weight gradients) --
        algorithmic FLOPs / summed launch durations, HIP events on the launch stream -- and the phases of the iteration."""
        torch.cuda.synchronize()
        ops.CONV_PROFILE, ops.CONV_BYTES, ops.CONV_KERNELS = [], None, None
        t0 = time.perf_counter()
        self.trainer.backward(self.batch)
        torch.cuda.synchronize()
        t_bwd = time.perf_counter() - t0
        rec, ops.CONV_PROFILE = ops.CONV_PROFILE, None
        dense = [r for r in rec if len(r) == 4]
        flops = sum(r[0] for r in dense)
        ms = sum(r[1].elapsed_time(r[2]) for r in dense)
        wg = [r for r in dense if r[3].startswith("wgrad")]
        wg_ms = sum(r[1].elapsed_time(r[2]) for r in wg)
        agg = {}
        for r in wg:
            a = agg.setdefault(r[3], [0, 0.0, 0.0])
            a[0] += 1
            a[1] += r[1].elapsed_time(r[2])
            a[2] += r[0]
        self._wgrad_top = [{"shape": k, "calls": v[0], "ms": round(v[1], 2), "tf": round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                           for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]]
        # phases of one iteration, each bracketed by a device synchronise
        ph = {}
        t0 = time.perf_counter()
        self.trainer.grads.all_reduce_mean()
        torch.cuda.synchronize()
        ph["all_reduce_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        self.trainer.opt.step(live_ranges=self.trainer.live_ranges)
        torch.cuda.synchronize()
        ph["clip_adamw_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        self.trainer._prepare()
        torch.cuda.synchronize()
        ph["prepare_operands_ms"] = (time.perf_counter() - t0) * 1e3
        ph["forward_backward_ms_profiled"] = t_bwd * 1e3
        self._phases = {k: round(v, 2) for k, v in ph.items()}
        self._phases["loss"] = float(self.last["loss"]) if self.last is not None else None
        self._phases["peak_memory_gb"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
        self._phases["batchnorm"] = "running statistics (frozen)" if self.frozen_bn else "batch statistics (model.train())"
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self._all_reduce = self.all_reduce_report()
        peak = MFMA_PEAK_TF[self.dtype]
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        self._roofline = {
            "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": None, "kernel": "all dense conv / linear launches of one iteration (sparse gathered GEMMs excluded)",
            "launches": len(dense), "kernel_ms": round(ms, 2), "gflop": round(flops / 1e9, 1),
            "wgrad": {"launches": len(wg), "ms": round(wg_ms, 2),
                      "tflops": round(sum(r[0] for r in wg) / max(wg_ms * 1e-3, 1e-9) / 1e12, 1),
                      "note": ("layers with >= 64 channels a side: LDS-staged bf16x3 kernel (three bf16 MFMAs per product); the "
                               "rest: exact-f32 MFMA (peak 157.3 TFLOP/s)") if self.dtype == "bf16x3"
                      else "exact-f32 MFMA (peak 157.3 TFLOP/s)"}}

    def roofline(self):
        return self._roofline

    def all_reduce_report(self, repeats=5):
        """The iteration's one data-path collective on its own (every rank calls this): the flat f32 gradient buffer through
        `FlatGradBuffer.all_reduce_mean` (RCCL over xGMI), `repeats` times, host-synchronised -- median time, bytes, and the
        bandwidths a ring prices: algbw = bytes / t; busbw = algbw x 2 (N-1) / N (what each GPU sends and receives); per-link =
        busbw / (N - 1) outgoing xGMI links in use on a fully connected node (7 links x ~153 GB/s per GPU at N = 8)."""
        import statistics
        import torch.distributed as dist
        g = self.trainer.grads
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        nbytes = g.flat.numel() * g.flat.element_size()
        ts = []
        sync = torch.cuda.synchronize if g.flat.is_cuda else (lambda: None)
        for _ in range(repeats):
            sync()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            g.all_reduce_mean()
            sync()
            ts.append(time.perf_counter() - t0)
        t = statistics.median(ts)
        rep = {"bytes": nbytes, "ms": round(t * 1e3, 3), "ms_all": [round(x * 1e3, 3) for x in ts], "world": world,
               "collective": "one all-reduce (SUM of pre-scaled f32) over the flat gradient buffer, RCCL"}
        if world > 1:
            alg = nbytes / t / 1e9
            bus = alg * 2 * (world - 1) / world
            rep.update({"algbw_gbs": round(alg, 1), "busbw_gbs": round(bus, 1), "per_link_gbs": round(bus / (world - 1), 1),
                        "xgmi_link_peak_gbs": 153.0, "per_link_frac": round(bus / (world - 1) / 153.0, 3)})
        return rep

    def extra(self):
        return {"train_step_phases": getattr(self, "_phases", None), "wgrad_top_shapes": getattr(self, "_wgrad_top", None),
                "all_reduce": getattr(self, "_all_reduce", None)}

    def cpu_baseline(self):
        return None
