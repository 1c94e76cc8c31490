"""Gradient exchange (bucketed all-reduce, grads as views of flat buckets) on 2 CPU ranks over gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    from cleanumamba_amd.training.train_distributed import (apply_gradient_allreduce, init_distributed,
                                                          reduce_tensor)
    torch.set_num_threads(1)
    init_distributed(rank, world, None, "gloo", f"tcp://127.0.0.1:{port}")
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    net = nn.Sequential(nn.Conv1d(1, 8, 4, 2), nn.ReLU(), nn.Conv1d(8, 16, 1), nn.Conv1d(16, 1, 3), nn.Linear(7, 3))
    unused = nn.Linear(3, 3)                            # never receives a gradient
    net.add_module("unused", unused)
    net = apply_gradient_allreduce(net, bucket_bytes=256)   # tiny buckets -> several collectives
    assert len(net.grad_buckets.buckets) > 2
    ref = [p.detach().clone() for p in net.parameters()]
    gathered = [torch.zeros_like(ref[0]) for _ in range(world)]
    dist.all_gather(gathered, ref[0])
    assert torch.equal(gathered[0], gathered[1]), "parameters were not broadcast from rank 0"

    for step in range(2):
        net.grad_buckets.zero_grad()
        g = torch.Generator().manual_seed(7 + 10 * step + rank)
        x = torch.randn(4, 1, 20, generator=g)
        y = net[4](net[3](net[2](net[1](net[0](x)))))
        local = torch.autograd.grad(y.square().mean(), [p for p in net.parameters() if p is not unused.weight
                                                       and p is not unused.bias], retain_graph=True)
        y.square().mean().backward()
        params = [p for p in net.parameters() if p is not unused.weight and p is not unused.bias]
        for p, lg in zip(params, local):
            both = [torch.zeros_like(lg) for _ in range(world)]
            dist.all_gather(both, lg.contiguous())
            want = (both[0] + both[1]) / world
            assert torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7), "averaged gradient mismatch"
            idx, view = net.grad_buckets.where[id(p)]
            assert p.grad.data_ptr() == view.data_ptr(), "grad is not a view of its bucket"
        assert torch.all(unused.weight.grad == 0)
    # the captured-step form of the exchange: backward WITHOUT the in-backward collectives (require_sync = False, what
    # graph 1 of TrainStep records), then one all-reduce of the whole flat gradient buffer -> the same averaged gradients
    want = net.grad_buckets.flat.grad.clone()
    net.grad_buckets.zero_grad()
    net.grad_buckets.require_sync = False
    y = net[4](net[3](net[2](net[1](net[0](x)))))
    y.square().mean().backward()
    assert not net.grad_buckets.handles and not net.grad_buckets._armed
    local_only = net.grad_buckets.flat.grad.clone()
    net.grad_buckets.exchange_all()
    net.grad_buckets.require_sync = True
    assert torch.allclose(net.grad_buckets.flat.grad, want, rtol=1e-6, atol=1e-7)
    assert not torch.allclose(local_only, want, rtol=1e-3, atol=1e-5), "ranks saw different data: local != averaged"
    loss = torch.tensor([float(rank + 1)])
    assert reduce_tensor(loss, world).item() == 1.5
    # the ranks' choice between the replayed and the eager step is collective: one dissenting rank decides for all
    assert net.grad_buckets.all_ranks_agree(True) is True
    assert net.grad_buckets.all_ranks_agree(rank == 0) is False
    assert net.grad_buckets.all_ranks_agree(False) is False
    out.put((rank, "ok"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, "ok"), (1, "ok")]


def _alone_worker(port, out):
    """One rank of an initialised group: without CUM_EXCHANGE_ALONE nothing is exchanged (world size 1); with it every
    collective of the step runs over the one rank and the gradients stay what they were."""
    from cleanumamba_amd.training.train_distributed import GradBuckets, init_distributed
    torch.set_num_threads(1)
    init_distributed(0, 1, None, "gloo", f"tcp://127.0.0.1:{port}")
    torch.manual_seed(3)
    x = torch.randn(4, 1, 20)
    grads = {}
    for alone in ("0", "1"):
        os.environ["CUM_EXCHANGE_ALONE"] = alone
        torch.manual_seed(4)
        net = nn.Sequential(nn.Conv1d(1, 8, 4, 2), nn.ReLU(), nn.Conv1d(8, 16, 1))
        buckets = GradBuckets(net, bucket_bytes=256)
        assert buckets.world == 1 and buckets.exchanging == (alone == "1")
        launched = []
        real = buckets._launch
        buckets._launch = lambda flat: (launched.append(flat.numel()), real(flat))[1]
        buckets.zero_grad()
        net(x).square().mean().backward()
        assert bool(launched) == (alone == "1") and not buckets.handles
        before = buckets.flat.grad.clone()
        buckets.exchange_all()
        assert torch.equal(buckets.flat.grad, before)
        grads[alone] = before
    assert torch.allclose(grads["0"], grads["1"], rtol=1e-6, atol=1e-8)
    out.put("ok")
    dist.destroy_process_group()


def test_one_rank_group_exchanges_only_on_request():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_alone_worker, args=(_free_port(), q))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"
