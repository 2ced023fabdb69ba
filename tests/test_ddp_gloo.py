# -*- coding: utf-8 -*-
"""N>1 path on CPU: world_size-2 gloo run of the bucketed gradient exchange
(yolov4_amd/ddp.py).  The wrapper is model agnostic, so a small torch module
stands in for the detector (whose kernels need a GPU)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1),
                               torch.nn.Flatten(), torch.nn.Linear(8 * 6 * 6, 10))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from yolov4_amd.ddp import BucketedDDP
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    net = _net()
    if rank != 0:                                   # wrong weights on rank 1: the wrap-time broadcast must fix them
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    ddp = BucketedDDP(net, bucket_mb=0.001)          # tiny buckets -> several exchanges in flight
    assert len(ddp.buckets) >= 3
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 3, 6, 6, generator=g)
    y = torch.randn(8, 10, generator=g)
    for step in range(2):                            # second step checks the zero_grad / re-arm logic
        ddp.zero_grad()
        xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
        loss = ((ddp(xs) - ys) ** 2).sum()           # summed (not averaged) loss, like YOLOLoss
        loss.backward()
        ddp.finish_backward()
    if rank == 0:
        torch.save([p.grad.clone().contiguous() for p in net.parameters()], out)
    # accumulation window of 2 micro-steps: exchange only on the second one; result = 2x the single-step grads
    single = [p.grad.clone() for p in net.parameters()]
    ddp.zero_grad()
    for micro in range(2):
        ddp.accumulating = micro == 0
        if micro:
            ddp.rearm()
        loss = ((ddp(xs) - ys) ** 2).sum()
        loss.backward()
        ddp.finish_backward()
    for a, p in zip(single, net.parameters()):
        torch.testing.assert_close(p.grad, 2 * a, rtol=1e-5, atol=1e-6)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_single_process(tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'grads.pt')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    net = _net()
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 3, 6, 6, generator=g)
    y = torch.randn(8, 10, generator=g)
    ((net(x) - y) ** 2).sum().backward()
    # sum over the 2 local batches / world_size (apex semantics) == full-batch gradient / 2
    for a, p in zip(got, net.parameters()):
        torch.testing.assert_close(a, p.grad / 2, rtol=1e-5, atol=1e-6)


def test_single_process_is_a_noop_wrapper():
    sys.path.insert(0, ROOT)
    from yolov4_amd.ddp import BucketedDDP
    net = _net()
    ddp = BucketedDDP(net, bucket_mb=0.001)
    x = torch.randn(2, 3, 6, 6)
    ddp(x).sum().backward()
    ddp.finish_backward()
    ref = _net()
    ref(x).sum().backward()
    for a, b in zip(net.parameters(), ref.parameters()):
        torch.testing.assert_close(a.grad, b.grad)
    w = net[0].weight
    assert w.grad.data_ptr() >= ddp.buckets[-1].flat.data_ptr()     # grads are views into the flat bucket
