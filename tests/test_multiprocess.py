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


# ------------------------------------------------------------------------------------------------- GPU collectives
import pytest  # noqa: E402


@pytest.mark.gpu
def test_rccl_single_rank_flat_gradient_all_reduce():
    """First RCCL execution of the training collective: a 1-rank backend="nccl" (= RCCL on ROCm) process
    group on this box's GPU, the flat gradient buffer's all-reduce (a no-op arithmetic-wise at world size 1 -- FlatGradBuffer
    short-cuts it, so the collective is also issued directly) and the packed scalar all-reduce of parse_losses."""
    from thinktwice_amd.grad_sync import FlatGradBuffer
    from thinktwice_amd.losses import parse_losses
    port = 32500 + (os.getpid() % 2000)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        params = [torch.randn(1000, 37, device=dev).requires_grad_(True), torch.randn(5, device=dev).requires_grad_(True)]
        buf = FlatGradBuffer(params)
        buf.flat.copy_(torch.arange(buf.numel, device=dev, dtype=torch.float32) / buf.numel)
        want = buf.flat.clone()
        assert buf.all_reduce_mean() is None                       # world size 1: short-cut
        dist.all_reduce(buf.flat, op=dist.ReduceOp.SUM)            # the collective itself, on RCCL
        stats = torch.full((2, 130), 3.0, dtype=torch.float64, device=dev)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)               # the SyncBN statistics block (f64)
        torch.cuda.synchronize()
        assert torch.equal(buf.flat, want) and bool((stats == 3.0).all())
        loss, log_vars = parse_losses({"a_loss": torch.tensor(1.5, device=dev), "b_offset": torch.tensor(2.0, device=dev)})
        assert abs(float(loss) - 1.5) < 1e-6 and abs(log_vars["b_offset"] - 2.0) < 1e-6
    finally:
        dist.destroy_process_group()


def _syncbn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from thinktwice_amd import autodiff, ops
        g = torch.Generator().manual_seed(77)
        C = 24
        z_all = torch.randn(2, 3, 5, 6, C, generator=g) * 2.0 + 1.0          # [rank][N,H,W,C]
        R_all = torch.randn(2, 3, 5, 6, C, generator=g)
        gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
        spec = ops.BNSpec("bn", gamma.cuda(), beta.cuda(), torch.zeros(C).cuda(), torch.ones(C).cuda(), 1e-5, 0.1)
        z = z_all[rank].cuda()
        with autodiff.Tape() as tape:
            y = ops.batchnorm_train(z, spec, 1)
            tape.seed(y, R_all[rank])
            tape.backward()
            dz = tape.grad(z).cpu()
        torch.cuda.synchronize()
        q.put((rank, y.cpu().numpy(), dz.numpy(), spec.running_mean.cpu().numpy(), spec.running_var.cpu().numpy(),
               tape.param_grads["bn.weight"].cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_syncbn_two_ranks_share_statistics_gloo():
    """SyncBN (configs/thinktwice.py:39, apis/mmdet_train.py:86-87): two ranks (both on this box's GPU, gloo) normalise their
    own half of a batch with the statistics of the WHOLE batch -- outputs, input gradients and running statistics equal
    torch's single-process BatchNorm over the concatenated batch; the running statistics are identical on both ranks."""
    import numpy as np
    import torch.nn.functional as F
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(77)
    C = 24
    z_all = (torch.randn(2, 3, 5, 6, C, generator=g) * 2.0 + 1.0).requires_grad_(True)
    R_all = torch.randn(2, 3, 5, 6, C, generator=g)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).requires_grad_(True), torch.randn(C, generator=g)
    rm, rv = torch.zeros(C), torch.ones(C)
    zz = z_all.reshape(6, 5, 6, C).permute(0, 3, 1, 2)
    y = torch.relu(F.batch_norm(zz, rm, rv, gamma, beta, True, 0.1, 1e-5)).permute(0, 2, 3, 1).reshape(2, 3, 5, 6, C)
    (y * R_all).sum().backward()
    for rank, yy, dz, grm, grv, dgamma in got:
        np.testing.assert_allclose(yy, y[rank].detach().numpy(), atol=2e-5)
        np.testing.assert_allclose(dz, z_all.grad[rank].numpy(), atol=2e-4 * float(z_all.grad.abs().max()))
        np.testing.assert_allclose(grm, rm.numpy(), atol=1e-6)
        np.testing.assert_allclose(grv, rv.numpy(), atol=1e-5)
    np.testing.assert_array_equal(got[0][3], got[1][3])
    np.testing.assert_array_equal(got[0][4], got[1][4])
    # dgamma is the LOCAL sum on each rank; the two add up to the whole batch's (the gradient all-reduce averages them)
    np.testing.assert_allclose(got[0][5] + got[1][5], gamma.grad.numpy(), atol=2e-4 * float(gamma.grad.abs().max()))


def _allreduce_report_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from thinktwice_amd.bench_train import TrainStepWorkload
    from thinktwice_amd.grad_sync import FlatGradBuffer
    p = torch.zeros(1 << 16).requires_grad_()
    buf = FlatGradBuffer([p])
    buf.flat.fill_(float(rank + 1))
    wl = TrainStepWorkload.__new__(TrainStepWorkload)          # the report only needs trainer.grads
    wl.trainer = type("T", (), {"grads": buf})()
    rep = wl.all_reduce_report(repeats=3)
    q.put((rank, rep, float(buf.flat[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_all_reduce_report_two_ranks_gloo():
    """The N-GPU bench record of the training collective (VERDICT r3 missing #3): every rank runs the report, it times the
    flat-gradient all-reduce and prices it as a ring (busbw = algbw x 2 (N-1) / N, per link = busbw / (N-1))."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_allreduce_report_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rep, v in out:
        assert rep["world"] == 2 and rep["bytes"] == 4 << 16 and len(rep["ms_all"]) == 3 and rep["ms"] > 0
        assert abs(rep["busbw_gbs"] - rep["algbw_gbs"]) <= 0.11           # 2 (N-1) / N = 1 at N = 2 (rounded to 0.1)
        assert abs(rep["per_link_gbs"] - rep["busbw_gbs"]) <= 0.11 and 0 <= rep["per_link_frac"]
        # three repeated means of (1, 2): 1.5 -> stays 1.5 on both ranks after the first (mean of equal values)
        assert abs(v - 1.5) < 1e-6
