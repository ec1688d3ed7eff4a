"""N>1 path on CPU: two gloo ranks run the bench timing harness (barrier, max-over-ranks, whole-job
rate) and the frame sharding rule.  Inference = replicas only, so no data-path collective exists to test."""
import os
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from thinktwice_amd.bench_harness import run_timed, shard_frames


class _FakeWorkload:
    def __init__(self, rank):
        self.rank = rank
        self.calls = 0

    def step(self):
        self.calls += 1
        time.sleep(0.02 * (1 + self.rank))     # rank 1 is the straggler

    def frames_per_step(self):
        return 8


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = _FakeWorkload(rank)
    res = run_timed(wl, steps=5, warmup=2, dist=dist)
    q.put((rank, res, wl.calls))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_harness_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, a, c0), (r1, b, c1) = out
    assert c0 == c1 == 7                                      # warm-up 2 + timed 5 on every rank
    assert a["world"] == b["world"] == 2
    assert abs(a["seconds"] - b["seconds"]) < 1e-9            # MAX over ranks is what everyone reports
    assert a["seconds"] >= 5 * 0.04 * 0.95                    # bounded by the straggler
    assert abs(a["value"] - 8 * 5 * 2 / a["seconds"]) < 1e-6  # whole-job frames / max time


def test_frame_sharding_is_a_partition():
    for total in (1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            spans = [shard_frames(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
