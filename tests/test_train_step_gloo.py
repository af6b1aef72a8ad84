"""CPU, 2 gloo ranks: the data-parallel step's host logic (flat-bucket all-reduce, grad binding, clipping, optimizer)
with a stand-in model whose backward fills a flat bucket the way the HIP backward does."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lfsr_amd.train_step import allreduce_bucket, bind_grads_to_bucket, broadcast_parameters, train_step


class FakeNet(torch.nn.Module):
    """y = a*x + b elementwise; 'backward' is done by autograd, then copied into a flat bucket like the HIP path."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.tensor([1.0, 2.0]))
        self.b = torch.nn.Parameter(torch.tensor([0.5]))
        self._spans = {"a": (0, 2), "b": (2, 1)}
        self.grad_bucket = torch.zeros(3)

    def forward(self, x, info=None):
        y = x * self.a.sum() + self.b
        if y.requires_grad:
            y.register_hook(lambda g: None)
        return y


def fill_bucket(net):
    net.grad_bucket = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(rank)                      # different init per rank ...
        net = FakeNet()
        with torch.no_grad():
            net.a.add_(rank)
        broadcast_parameters(net)                    # ... made identical
        ok = torch.equal(net.a.data, torch.tensor([1.0, 2.0]))
        x = torch.full((4,), float(rank + 1))        # each rank its own shard
        y = net(x)
        loss = (y - 1.0).abs().mean()
        loss.backward()
        fill_bucket(net)
        local = net.grad_bucket.clone()
        allreduce_bucket(net.grad_bucket)
        gathered = [torch.zeros(3) for _ in range(world)]
        dist.all_gather(gathered, local)
        ok = ok and torch.allclose(net.grad_bucket, sum(gathered) / world)
        bind_grads_to_bucket(net)
        ok = ok and net.a.grad.data_ptr() == net.grad_bucket.data_ptr()          # views, not copies
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        pa = [torch.zeros(2) for _ in range(world)]
        dist.all_gather(pa, net.a.data)
        ok = ok and torch.equal(pa[0], pa[1])                                     # replicas stay identical
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_bucket_allreduce():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(0) is True and ret.get(1) is True


def test_single_process_is_a_noop():
    b = torch.tensor([1.0, 2.0])
    assert allreduce_bucket(b) is b and torch.equal(b, torch.tensor([1.0, 2.0]))
