"""GPU study (tools only): output error vs the F14 golden for mixed precision modes -- which module may run in a 16-bit
storage mode while the rest stays bf16x3.  python tools/precision_mix_gpu.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from thinktwice_amd import model as tm, params, synth  # noqa: E402
import test_forward as TF  # noqa: E402

pack = np.load(os.path.join(ROOT, "tests", "golden", "f14_forward_full_b8.npz"))
B, H, W, npts, seed = (int(v) for v in pack["meta"])
batch = tm.batch_to_device(synth.make_batch(B, img_hw=(H, W), num_points=npts))
for name, kw in (("x3 all", dict(dtype="f32x3")), ("x3 + lidar f16", dict(dtype="f32x3", lidar_dtype=torch.float16)),
                 ("x3 + lidar bf16", dict(dtype="f32x3", lidar_dtype=torch.bfloat16))):
    m, cfg = tm.build_thinktwice(final_dim=(H, W), **kw)
    m.load_state_dict(params.init_params(cfg, seed=seed))
    out = m.forward_inference(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.forward_inference(batch, channel_last_out=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    try:
        errs = TF._check_against_pack(pack, out, 1.0)
    except AssertionError as e:
        errs = {"assert": str(e)[:200]}
    inter = TF._inter_errs(pack, out)
    worst = max(errs.values()) if "assert" not in errs else None
    print(f"{name:18s} {dt * 1e3:7.1f} ms/step  worst {worst}  pred_wp {errs.get('pred_wp')}  value_traj {errs.get('pred_value_traj')} "
          f"inter {inter}", flush=True)
    del m
    torch.cuda.empty_cache()
