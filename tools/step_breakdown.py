#!/usr/bin/env python
"""Where the batch-8 step goes: the forward's sections timed serially on one stream (LiDAR encoder, camera encoder, fusion +
decoder), the whole forward serial vs with the LiDAR side stream, and the forward with the LiDAR branch's result CACHED (not a
valid forward: the upper bound of what a faster sparse encoder could give)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import bench_forward, ops  # noqa: E402


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    wl = bench_forward.ForwardWorkload(B, torch.device("cuda", 0))
    m, batch = wl.model, wl.batch
    dev = m.device
    print(f"B={B} whole forward, side stream on : {timed(wl.step):.2f} ms")
    m.use_side_stream = False
    print(f"B={B} whole forward, one stream     : {timed(wl.step):.2f} ms")
    pts = batch["points"][:, -1].to(dev)
    print(f"  LiDAR encoder alone              : {timed(lambda: m.lidar_encoder(pts, channel_last=True, rot_flip=True)):.2f} ms")
    img = batch["img"].to(dev)
    print(f"  camera encoder alone             : {timed(lambda: m.img_encoder(img, batch['img_metas'], channel_last=True)):.2f} ms")
    # camera trunk pieces
    enc = m.img_encoder
    lidar = m.lidar_encoder(pts, channel_last=True, rot_flip=True)
    m.use_side_stream = True
    real = m.lidar_encoder

    class Cached:
        def __call__(self, *a, **k):
            return lidar
    m.lidar_encoder = Cached()
    print(f"B={B} forward with the LiDAR result cached (NOT a valid forward): {timed(wl.step):.2f} ms")
    m.lidar_encoder = real
    # conv-level table of the serial forward, by kernel family
    ops.CONV_PROFILE, ops.CONV_KERNELS = [], []
    m.use_side_stream = False
    wl.step()
    torch.cuda.synchronize()
    fam = {}
    for r, k in zip(ops.CONV_PROFILE, ops.CONV_KERNELS):
        ms = r[1].elapsed_time(r[2])
        key = k.split("<")[0] + ("<" + k.split("<")[1][:28] if "<" in k else "")
        fam[key] = fam.get(key, 0.0) + ms
    ops.CONV_PROFILE = ops.CONV_KERNELS = None
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f"    {v:8.3f} ms  {k}")
    print(f"    {sum(fam.values()):8.3f} ms  all conv launches (serial)")


if __name__ == "__main__":
    main()
