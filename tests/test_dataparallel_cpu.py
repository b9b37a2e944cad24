"""Data-parallel host logic on CPU (gloo, world_size 2): the pieces of scripts/train.py:100-102 semantics that
the HIP engine restates explicitly (fsnet_amd/engine/dataparallel.py) and the rank-strided sampler.
The device kernels are not involved: the exchanged buffers are plain tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(fn, world=2):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    for r in res:
        assert r[1] is None, "rank %d failed: %s" % (r[0], r[1])
    return dict((r[0], r[2]) for r in res)


def _entry(fn, rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        out = fn(rank, world)
        dist.destroy_process_group()
        q.put((rank, None, out))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None))


# ---------------------------------------------------------------------------------------------------
def _syncbn_case(rank, world):
    """SyncBN statistics exchange: per-rank slot buffers (sum, sumsq) all-reduced == global-batch BN."""
    from fsnet_amd.engine.dataparallel import DataParallelContext
    torch.manual_seed(0)
    x_all = torch.randn(world * 3, 8, 5, 7) * 2 + 1
    x = x_all[rank * 3:(rank + 1) * 3]
    dp = DataParallelContext(meta_arch=None)
    stats = torch.zeros(8, 2, 8, dtype=torch.float64)   # [FS_STAT_SLOTS][2][C]: what the conv epilogue fills
    stats[rank % 8, 0] = x.double().sum(dim=(0, 2, 3))
    stats[(rank + 3) % 8, 1] = (x.double() ** 2).sum(dim=(0, 2, 3))
    dp.allreduce_small(stats)
    count = x.shape[0] * 5 * 7 * dp.world
    s = stats.sum(0)
    mean = s[0] / count
    var = s[1] / count - mean ** 2
    ref = F.batch_norm(x_all, None, None, training=True)
    got = (x_all - mean.float().view(1, -1, 1, 1)) / torch.sqrt(var.float().view(1, -1, 1, 1) + 1e-5)
    return float((got - ref).abs().max())


def test_syncbn_statistics_exchange():
    res = _run(_syncbn_case)
    assert all(v < 1e-5 for v in res.values())


def _grad_bucket_case(rank, world):
    """flat-arena gradient all-reduce per network + SUM->MEAN scale == DDP's averaged gradient."""
    import torch.nn as nn
    from fsnet_amd.engine.dataparallel import DataParallelContext
    from fsnet_amd.engine.runtime import ParamArena

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4))
            self.b = nn.Linear(5, 7)
    torch.manual_seed(1)
    net = Net()
    arena = ParamArena(list(net.named_parameters()), torch.device("cpu"))
    net._arena = arena
    assert arena.intact()
    dp = DataParallelContext(net)
    # rank-dependent parameters / buffers -> begin_step broadcasts rank 0's
    with torch.no_grad():
        arena.data.add_(float(rank))
        net.a[1].running_mean.add_(float(rank))
    dp.begin_step(net)
    p_after = arena.data.clone()
    arena.zero_grads()
    torch.manual_seed(100 + rank)
    for p in net.parameters():
        p.grad.copy_(torch.randn_like(p))
    local = arena.grad.clone()
    dp.grads_ready(net.b)
    dp.grads_ready(net.a)
    scale = dp.finish()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered) / world
    err = float((arena.grad * scale - expect).abs().max())
    return err, float(p_after.sum()), float(net.a[1].running_mean.sum()), [float(p.grad.sum()) for p in net.parameters()][:2]


def _partial_bucket_case(rank, world):
    """stage-level buckets: partial_ready() reduces a child's slice early, grads_ready() the rest of the network —
    every element reduced exactly once; finish() raises when a network that ran a forward was never reduced, and
    begin_step() resets pending counters a forward without backward left behind."""
    import torch.nn as nn
    from fsnet_amd.engine.dataparallel import DataParallelContext
    from fsnet_amd.engine.runtime import ParamArena

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 4, 3)
            self.layer1 = nn.Sequential(nn.Conv2d(4, 4, 3), nn.BatchNorm2d(4))
            self.layer2 = nn.Sequential(nn.Conv2d(4, 8, 3), nn.BatchNorm2d(8))
            self._pending = 0

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.enc = Enc()
            self.head = nn.Linear(5, 7)
            self.head._pending = 0
    torch.manual_seed(2)
    net = Net()
    arena = ParamArena(list(net.named_parameters()), torch.device("cpu"))
    net._arena = arena
    dp = DataParallelContext(net)
    net.enc._pending = 3                      # a forward that never saw its backward
    dp.begin_step(net)
    assert net.enc._pending == 0
    arena.zero_grads()
    torch.manual_seed(200 + rank)
    for p in net.parameters():
        p.grad.copy_(torch.randn_like(p))
    local = arena.grad.clone()
    dp.note_forward(net.enc); dp.note_forward(net.head)
    dp.partial_ready(net.enc, [net.enc.layer2])       # reverse parameter order, like DDP's buckets
    dp.grads_ready(net.head)
    raised = False
    try:
        dp.finish()                                    # enc's conv1 / layer1 not reduced yet
    except RuntimeError:
        raised = True
    dp.grads_ready(net.enc)
    scale = dp.finish()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered) / world
    return float((arena.grad * scale - expect).abs().max()), raised


def test_stage_buckets_reduce_every_gradient_once_and_finish_checks_coverage():
    res = _run(_partial_bucket_case)
    assert all(v[0] < 1e-6 and v[1] for v in res.values()), res


def test_gradient_buckets_and_initial_broadcast():
    res = _run(_grad_bucket_case)
    assert all(v[0] < 1e-6 for v in res.values())
    assert abs(res[0][1] - res[1][1]) < 1e-6       # parameters identical after the rank-0 broadcast
    assert abs(res[0][2] - res[1][2]) < 1e-6       # buffers too
    assert res[0][3] == res[1][3]                  # param.grad views see the reduced arena


def _sampler_case(rank, world):
    from fsnet_amd.vision_base.data.dataloader.distributed_sampler import TrainingSampler
    s = TrainingSampler(103, rank=rank, world_size=world)
    return list(iter(s))


def test_training_sampler_is_rank_strided_shared_permutation():
    """reference distributed_sampler.py:48-56: same permutation on every rank, rank r keeps perm[r::world]."""
    res = _run(_sampler_case)
    g = torch.Generator()
    perm = torch.randperm(103, generator=g).tolist()
    assert res[0] == perm[0::2] and res[1] == perm[1::2]
    assert sorted(res[0] + res[1]) == list(range(103))


def _syncbn_train_case(rank, world):
    """conv -> BN(train) -> ReLU -> masked-mean loss on B=1 per rank, with the engine's exchange protocol:
    forward (sum, sumsq) all-reduced; backward (sum g, sum g*xhat) all-reduced for dx, LOCAL sums for
    dgamma/dbeta (fs_bn_bwd_apply), parameter gradients all-reduced and averaged.  Must equal single-process
    autograd on the concatenated batch (what SyncBatchNorm + DDP give the reference, scripts/train.py:100-102)."""
    from fsnet_amd.engine.dataparallel import DataParallelContext
    torch.manual_seed(5)
    X = torch.randn(world, 3, 9, 11)
    Wt = (torch.randn(6, 3, 3, 3) * 0.3)
    gam = 1 + 0.2 * torch.randn(6)
    bet = 0.1 * torch.randn(6)
    tgt = torch.randn(world, 6, 9, 11)
    # ---- single-process reference on the whole batch ----
    w_r, g_r, b_r = Wt.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y = F.relu(F.batch_norm(F.conv2d(X, w_r, padding=1), None, None, g_r, b_r, training=True, eps=1e-5))
    loss_ref = sum(((y[r:r + 1] - tgt[r:r + 1]) ** 2).mean() for r in range(world)) / world   # mean of per-rank means
    loss_ref.backward()
    # ---- this rank ----
    dp = DataParallelContext(meta_arch=None)
    x = X[rank:rank + 1]
    c = F.conv2d(x, Wt, padding=1)
    stats = torch.zeros(8, 2, 6, dtype=torch.float64)
    stats[rank % 8, 0] = c.double().sum(dim=(0, 2, 3)); stats[rank % 8, 1] = (c.double() ** 2).sum(dim=(0, 2, 3))
    dp.allreduce_small(stats)
    count = c.shape[0] * 9 * 11 * dp.world
    s = stats.sum(0)
    mean = (s[0] / count); var = (s[1] / count - mean ** 2).clamp_min(0)
    invstd = (1.0 / torch.sqrt(var + 1e-5)).float(); mean = mean.float()
    xhat = (c - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    yb = F.relu(xhat * gam.view(1, -1, 1, 1) + bet.view(1, -1, 1, 1))
    dy = 2 * (yb - tgt[rank:rank + 1]) / yb.numel()           # d(local mean loss)/dy ; DDP averages over ranks later
    g = dy * (yb > 0)
    sums = torch.zeros(8, 2, 6, dtype=torch.float64)
    sums[0, 0] = g.double().sum(dim=(0, 2, 3)); sums[0, 1] = (g * xhat).double().sum(dim=(0, 2, 3))
    local = sums.clone()
    dp.allreduce_small(sums)
    sg, sgx = sums.sum(0)[0].float() / count, sums.sum(0)[1].float() / count
    dc = (gam * invstd).view(1, -1, 1, 1) * (g - sg.view(1, -1, 1, 1) - xhat * sgx.view(1, -1, 1, 1))
    dgam, dbet = local.sum(0)[1].float(), local.sum(0)[0].float()
    dw = torch.nn.grad.conv2d_weight(x, Wt.shape, dc, padding=1)
    flat = torch.cat([dw.flatten(), dgam, dbet])
    dist.all_reduce(flat)
    flat = flat / world
    ref = torch.cat([w_r.grad.flatten(), g_r.grad, b_r.grad])
    return float((flat - ref).abs().max() / ref.abs().max())


def test_syncbn_backward_protocol_equals_big_batch_autograd():
    res = _run(_syncbn_train_case)
    assert all(v < 2e-5 for v in res.values()), res


def _agreement_case(rank, world):
    """the ranks' yes/no agreement and the small blob hand-off run through the process group's store: no collective,
    nothing the NCCL watchdog could poll (rccl_direct.StoreAgreement; used for the captured-step decision, the RCCL unique
    id and the communicator self-tests)"""
    from fsnet_amd.engine.dataparallel import DataParallelContext
    from fsnet_amd.engine.rccl_direct import StoreAgreement
    ag = StoreAgreement()
    out = [ag.all_agree(True), ag.all_agree(rank != 1), ag.all_agree(rank != 0), ag.all_agree(True)]
    blob = ag.share(b"unique-id-from-rank-0" if rank == 0 else b"ignored")
    dp = DataParallelContext(meta_arch=None)
    out += [dp.all_agree(True), dp.all_agree(rank == 0), dp.all_agree(True)]
    return out, blob


def test_rank_agreement_through_the_store():
    res = _run(_agreement_case)
    for r in (0, 1):
        out, blob = res[r]
        assert out == [True, False, False, True, True, False, True]
        assert blob == b"unique-id-from-rank-0"


def test_hand_overs_left_by_a_backward_that_raised_are_dropped_by_the_next_pass():
    """nets._LATE / _AT_END / _DEFERRED hold work that was put off inside a backward pass (flush_deferred, late_call); their
    end-of-backward callback empties them.  A pass that raised never ran it: the next pass must not issue that work (events of a
    dead capture, gradients of another step) — the first entry point it reaches drops it (nets._drop_stale)."""
    from fsnet_amd.engine import nets

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            nets._drop_stale()
            Probe.seen = (len(nets._LATE), len(nets._AT_END), sum(len(v[1]) for v in nets._DEFERRED.values()))
            return g * 2

    saved = (list(nets._LATE), list(nets._AT_END), nets._CALLBACK_QUEUED[0])
    try:
        nets._LATE.append(("event", None, None, None, 0))
        nets._AT_END.append(("event", None, []))
        nets._DEFERRED[-1] = (None, ["item"])
        nets._CALLBACK_QUEUED[0] = -12345                      # a pass whose callback never ran
        x = torch.ones(3, requires_grad=True)
        Probe.apply(x).sum().backward()
        assert Probe.seen == (0, 0, 0)
        # ... while the running pass's own entries stay
        nets._LATE.append(("event", None, None, None, 0))
        nets._CALLBACK_QUEUED[0] = None
        Probe.apply(x).sum().backward()
        assert Probe.seen[0] == 1
    finally:
        nets._DEFERRED.pop(-1, None)
        nets._LATE[:] = saved[0]
        nets._AT_END[:] = saved[1]
        nets._CALLBACK_QUEUED[0] = saved[2]
