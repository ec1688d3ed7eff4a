#!/usr/bin/env python
"""Where do the forward's device-to-device copies come from?  Runs one batch-8 forward with Tensor.copy_ / clone / contiguous /
to / repeat wrapped and prints the call sites (thinktwice_amd/*.py file:line) by count.  (rocprofv3 lists ~630
__amd_rocclr_copyBuffer launches per forward.)"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import model as tm, params, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m, cfg = tm.build_thinktwice(dtype="f32x3")
m.load_state_dict(params.init_params(cfg, seed=0))
batch = tm.batch_to_device(synth.make_batch(B))
m.forward_inference(batch, channel_last_out=True)
torch.cuda.synchronize()
sites = collections.Counter()
nbytes = collections.Counter()


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if self.is_cuda or (torch.is_tensor(out) and out.is_cuda):
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "thinktwice_amd" in fr.filename:
                    key = f"{name:10s} {os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70]}"
                    sites[key] += 1
                    nbytes[key] += self.numel() * self.element_size()
                    break
        return out
    setattr(torch.Tensor, name, f)


for n in ("copy_", "clone", "contiguous", "to", "repeat", "zero_", "fill_"):
    wrap(n)
for fn in ("zeros", "full", "cat", "stack"):
    orig = getattr(torch, fn)

    def g(*a, _orig=orig, _fn=fn, **k):
        out = _orig(*a, **k)
        if torch.is_tensor(out) and out.is_cuda:
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "thinktwice_amd" in fr.filename:
                    key = f"torch.{_fn:7s} {os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70]}"
                    sites[key] += 1
                    nbytes[key] += out.numel() * out.element_size()
                    break
        return out
    setattr(torch, fn, g)
m.forward_inference(batch, channel_last_out=True)
torch.cuda.synchronize()
print(f"B={B}: {sum(sites.values())} wrapped tensor ops on device tensors in one forward")
for k, v in sites.most_common(60):
    print(f"{v:5d}  {nbytes[k] / 1e6:10.2f} MB  {k}")
