"""CPU, world_size 2 over gloo: the data-parallel host logic -- SyncBN statistics exchange and the
flat-gradient all-reduce -- give the single-process (global batch) result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rcf_amd  # noqa
    from rcf_amd.layers import DistCtx
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 8, 5, 6, generator=g, dtype=torch.float64) * 2 + 1      # global batch, NCHW
    mine = x[rank * 2:(rank + 1) * 2]
    ctx = DistCtx()
    assert ctx.on and ctx.world == world
    # what rcf_bn_stats_f32 produces locally: [sum | sum of squares] per channel, fp64
    sums = torch.cat([mine.sum(dim=(0, 2, 3)), (mine * mine).sum(dim=(0, 2, 3))])
    ctx.allreduce_sum(sums)
    count = mine.numel() // 8 * world
    mean = sums[:8] / count
    var = sums[8:] / count - mean * mean
    # backward sums [sum g | sum g*xhat] likewise
    gy = torch.randn(4, 8, 5, 6, generator=g, dtype=torch.float64)
    xh = (mine - mean.view(1, 8, 1, 1)) / torch.sqrt(var.view(1, 8, 1, 1) + 1e-5)
    s2 = torch.cat([gy[rank * 2:(rank + 1) * 2].sum(dim=(0, 2, 3)), (gy[rank * 2:(rank + 1) * 2] * xh).sum(dim=(0, 2, 3))])
    ctx.allreduce_sum(s2)
    # two independent statistics vectors in ONE collective (a bottleneck's conv1 / downsample pair)
    a, b = torch.full((6,), float(rank + 1), dtype=torch.float64), torch.arange(4, dtype=torch.float64) * (rank + 1)
    n0 = ctx.count
    ctx.allreduce_sum_many([a, b])
    assert ctx.count == n0 + 1 and torch.equal(a, torch.full((6,), 3.0, dtype=torch.float64))
    assert torch.equal(b, torch.arange(4, dtype=torch.float64) * 3)
    # flat gradient all-reduce + 1/world scaling == gradient of the mean loss over the global batch
    grad = torch.full((1000,), float(rank + 1))
    dist.all_reduce(grad)
    q.put((rank, mean.numpy(), var.numpy(), s2.numpy(), float((grad / world)[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_syncbn_and_grad_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 8, 5, 6, generator=g, dtype=torch.float64) * 2 + 1
    gy = torch.randn(4, 8, 5, 6, generator=g, dtype=torch.float64)
    mean, var = x.mean(dim=(0, 2, 3)), x.var(dim=(0, 2, 3), unbiased=False)
    xh = (x - mean.view(1, 8, 1, 1)) / torch.sqrt(var.view(1, 8, 1, 1) + 1e-5)
    want_s2 = torch.cat([gy.sum(dim=(0, 2, 3)), (gy * xh).sum(dim=(0, 2, 3))]).numpy()
    for rank, m, v, s2, gavg in res:
        assert np.allclose(m, mean.numpy(), atol=1e-12) and np.allclose(v, var.numpy(), atol=1e-12)
        assert np.allclose(s2, want_s2, atol=1e-10)
        assert gavg == 1.5


def _group_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rcf_amd  # noqa
    from rcf_amd import trainer
    out = []
    g, mode = trainer.make_grad_group("auto", "cpu")                  # own communicator: created and probed
    t = torch.full((3,), float(rank + 1))
    dist.all_reduce(t, group=g)
    out.append((mode, g is not None, float(t[0])))
    out.append(trainer.make_grad_group(False, "cpu")[1])
    # ONE rank fails to create its communicator: every rank must fall back to the shared one ("auto") or raise (True)
    real = dist.new_group

    def failing(*a, **k):
        grp = real(*a, **k)                                           # stay collective, then fail locally
        if rank == 1:
            raise RuntimeError("simulated RCCL failure")
        return grp
    trainer.dist.new_group = failing
    try:
        g2, mode2 = trainer.make_grad_group("auto", "cpu")
        out.append((mode2, g2 is None))
        try:
            trainer.make_grad_group(True, "cpu")
            out.append("no error")
        except RuntimeError as e:
            out.append("raised: " + str(e)[:40])
    finally:
        trainer.dist.new_group = real
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_communicator_auto_falls_back_collectively():
    """SCHED.grad_group = "auto" (trainer.make_grad_group): own communicator when every rank can create and use one; when ANY
    rank fails, ALL ranks share the default communicator (the decision is an all-reduce, so no rank is left waiting on a group
    its peers gave up on); True raises on every rank instead."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_group_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        own, shared, fell, forced = res[rank]
        assert own == ("own communicator", True, 3.0)
        assert shared.startswith("shared with SyncBN")
        assert fell[1] is True and fell[0].startswith("shared with SyncBN (own communicator failed")
        assert forced.startswith("raised: SCHED.grad_group = True")
    assert "simulated RCCL failure" in res[1][2][0] and "on another rank" in res[0][2][0]
