#!/usr/bin/env python
"""A/B of the bench step with 1 / 2 / 3 batches in flight (alternating streams): ms per step, and the outputs of the pipelined
forwards against the serial one (bit-equal: the forward has no floating-point atomics)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import bench_forward  # noqa: E402

KEYS = ("pred_wp", "pred_value_traj", "pred_speed", "refine_future_BEV_feature", "mu_branches")


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    wl = bench_forward.ForwardWorkload(B, torch.device("cuda", 0))
    wl.pipeline = 1
    for _ in range(3):
        ref = wl.step()
    torch.cuda.synchronize()
    keys = [k for k, v in ref.items() if torch.is_tensor(v) and v.is_floating_point() and not k.startswith("_")]
    print("compared keys:", keys)
    ref = {k: ref[k].clone() for k in keys}
    for depth in (1, 2, 3, 1, 2):
        wl.pipeline, wl._streams, wl._tick = depth, None, 0
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [wl.step() for _ in range(steps)]
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        same = all(torch.equal(o[k], ref[k]) for o in outs[-3:] for k in keys)
        print(f"batches in flight {depth}: {ms:.2f} ms per step = {B / ms * 1e3:.2f} frames/s, outputs bit-equal to serial: {same}", flush=True)
        del outs
    print("faults", wl.model and __import__("thinktwice_amd.ops", fromlist=["x"]).chain_faults())


if __name__ == "__main__":
    main()
