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


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from thinktwice_amd.grad_sync import FlatGradBuffer
    g = torch.Generator().manual_seed(0)                        # same parameters on every rank
    params = [torch.randn(s, generator=g).requires_grad_() for s in ((5, 3), (7,), (2, 2, 2))]
    dead = torch.randn(4, generator=g).requires_grad_()         # never reached by the loss (reference: 90 such)
    buf = FlatGradBuffer(params + [dead])
    buf.zero_()
    x = torch.full((3,), float(rank + 1))                       # rank-dependent data
    loss = (params[0] @ x).sum() * (rank + 1) + (params[1] ** 2).sum() + params[2].sum() * rank
    loss.backward()
    assert all(p.grad.data_ptr() == buf.flat[o:o + p.numel()].data_ptr() for p, o in zip(buf.params, buf.offsets))
    local = buf.flat.clone()
    buf.all_reduce_mean()
    norm = buf.clip_grad_norm_(1e9)
    q.put((rank, local.numpy(), buf.flat.clone().numpy(), float(norm)))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_all_reduce_two_ranks_gloo():
    """SURVEY 8e training collective: one all-reduce (mean) over the flat gradient buffer; autograd accumulates
    straight into the buffer's views, dead parameters stay zero, both ranks end with the average of the local grads."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, a0, n0), (_, l1, a1, n1) = out
    import numpy as np
    np.testing.assert_allclose(a0, (l0 + l1) / 2, rtol=1e-6)
    np.testing.assert_array_equal(a0, a1)
    assert np.all(a0[-4:] == 0.0) and not np.allclose(l0, l1)   # dead parameter: zeros; the ranks really differed
    assert abs(n0 - float(np.linalg.norm(a0))) < 1e-4 and n0 == n1


def _loss_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from thinktwice_amd.losses import parse_losses
    losses = {"wp_loss": torch.tensor(1.0 + rank), "value_loss": torch.tensor([[0.5], [1.5 + rank]]),
              "lateral_offset": torch.tensor(10.0 * (rank + 1)), "aux": [torch.tensor(2.0), torch.tensor([4.0, 6.0])]}
    loss, log_vars = parse_losses(losses)
    q.put((rank, float(loss), dict(log_vars)))
    dist.barrier()
    dist.destroy_process_group()


def test_parse_losses_two_ranks_gloo():
    """EncoderDecoder._parse_losses (encoder_decoder_framework.py:409-439) under torch.distributed: the returned loss is
    the LOCAL sum of the entries whose name contains 'loss' (each reduced by its mean; lists summed), the logged values
    are averaged over the ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_loss_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, v0), (_, l1, v1) = out
    assert abs(l0 - (1.0 + 1.0)) < 1e-6 and abs(l1 - (2.0 + 1.5)) < 1e-6          # wp_loss + mean(value_loss); local
    assert v0 == v1                                                               # logged values are rank means
    assert abs(v0["wp_loss"] - 1.5) < 1e-6 and abs(v0["value_loss"] - 1.25) < 1e-6
    assert abs(v0["lateral_offset"] - 15.0) < 1e-6 and abs(v0["aux"] - 7.0) < 1e-6
    assert abs(v0["loss"] - 2.75) < 1e-6 and list(v0) == ["wp_loss", "value_loss", "lateral_offset", "aux", "loss"]
