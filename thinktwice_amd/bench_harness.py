"""Timing harness shared by bench.py and the multi-process tests: W untimed warm-up steps, then K
steps bracketed by barrier + device synchronise on both sides, MAX over ranks, whole-job frames/s.
Inference shards by frame -- replicas only, no data-path collective."""
import time


def shard_frames(global_frames, rank, world):
    """Contiguous shard of `global_frames` frames for `rank` (sizes differ by at most 1)."""
    base, rem = divmod(global_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def run_timed(workload, steps, warmup, dist=None, sync=lambda: None, device=None):
    import torch
    for _ in range(warmup):
        workload.step()
    if hasattr(workload, "on_warm"):
        workload.on_warm()
    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        workload.step()
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    world = 1
    if dist is not None:
        world = dist.get_world_size()
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frames = workload.frames_per_step() * steps * world
    return {"value": frames / dt, "ms_per_step": dt / steps * 1e3, "world": world, "seconds": dt}
