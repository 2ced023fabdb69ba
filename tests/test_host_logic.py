# -*- coding: utf-8 -*-
"""CPU-side checks of the host logic added in round 2 (no GPU, no kernels): the autograd context of ConvBNActFn must
not keep its own output alive, the drop-in DDP wrapper heals detached gradient slots and finishes by itself, the
optimizer's zero_grad keeps the slots attached, and the train/validate harness follows the reference's sequence."""
import gc
import os
import socket
import sys
import weakref

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import recipe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ reference cycle
def test_convbnact_ctx_does_not_hold_its_output(monkeypatch):
    """The pattern that leaked 4.96 GiB per step: forward returned the very tensor it had been handed as destination
    while ctx.cfg still referenced it.  With kernels mocked by torch CPU stand-ins, the output of a ConvBNAct that
    writes into a caller slot must die on `del` with the garbage collector OFF."""
    from yolov4_amd import ops
    from yolov4_amd.darknet.darknet import ConvBNAct

    def fake_bnstats(x, w, k, s, rm, rv, nbt, mom, eps, x_amax=None, dgrad_filter=None):
        y = torch.nn.functional.conv2d(x, w, None, s, (k - 1) // 2)
        return y, y.mean((0, 2, 3)), (y.var((0, 2, 3), unbiased=False) + eps).rsqrt()

    def fake_bn_act(y, mean, invstd, gamma, beta, act, residual=None, out=None, out_amax=None, **kw):
        z = (y - mean.view(1, -1, 1, 1)) * (invstd * gamma).view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        if out is not None:
            out.copy_(z)
            return out
        return z

    monkeypatch.setattr(ops, '_require_gpu', lambda t, what: None)
    monkeypatch.setattr(ops, 'f16x2_mode', lambda: False)      # no operand maxima (device kernels) in this host-logic test
    monkeypatch.setattr(ops, 'planes_mode', lambda: None)
    monkeypatch.setattr(ops, 'conv_fwd_bnstats_raw', fake_bnstats)
    monkeypatch.setattr(ops, 'bn_act_fwd_raw', fake_bn_act)
    m = ConvBNAct(8, 8, 1, 1).train()
    x = torch.randn(2, 8, 4, 4, requires_grad=True)
    gc.collect()
    gc.disable()
    try:
        buf = ops.CatBuffer(2, [8, 8], 4, 4, torch.device('cpu'))
        z = m(x, out=buf.slot(1))
        assert z.data_ptr() == buf.slot(1).data_ptr() and z.grad_fn is not None
        ref_z, ref_buf = weakref.ref(z), weakref.ref(buf.buf)
        del z, buf
        assert ref_z() is None, 'ConvBNAct output is kept alive by its own autograd context (reference cycle)'
        assert ref_buf() is None, 'the concat buffer outlives its last user'
    finally:
        gc.enable()


# ------------------------------------------------------------------ DDP drop-in behaviour
def _net():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 1),
                              torch.nn.Flatten(), torch.nn.Linear(8 * 6 * 6, 10))
    # KRSC filters as in the detector (channels_last 4-D parameters) and an odd-sized bias (slot alignment)
    for mod in net:
        if isinstance(mod, torch.nn.Conv2d):
            mod.weight.data = mod.weight.data.contiguous(memory_format=torch.channels_last)
    net[4] = torch.nn.Linear(8 * 6 * 6, 7)
    return net


def test_ddp_heals_slots_after_optimizer_zero_grad_and_finishes_by_itself():
    from yolov4_amd.ddp import BucketedDDP
    net = _net()
    ddp = BucketedDDP(net, bucket_mb=0.001)
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    x = torch.randn(4, 3, 6, 6)
    ref = _net()
    ref(x).sum().backward()
    for variant in range(3):
        if variant == 0:
            opt.zero_grad()                      # torch default set_to_none=True: every slot detached
            assert all(p.grad is None for p in net.parameters())
        elif variant == 1:
            opt.zero_grad(set_to_none=False)
        else:
            for p in net.parameters():           # someone replaces gradients by foreign tensors
                p.grad = torch.ones_like(p) * 3.0
        ddp(x).sum().backward()                  # no zero_grad()/finish_backward() calls around it
        assert ddp._finished and all(b.pending == 0 and b.launched for b in ddp.buckets)
        for b in ddp.buckets:
            lo, hi = b.flat.data_ptr(), b.flat.data_ptr() + b.flat.numel() * 4
            for p, v in zip(b.params, b.views):
                assert lo <= p.grad.data_ptr() < hi and p.grad.data_ptr() % 16 == 0
                assert p.grad.data_ptr() == v.data_ptr() and p.grad.stride() == v.stride()
        for a, b in zip(net.parameters(), ref.parameters()):
            torch.testing.assert_close(a.grad, b.grad)
    assert ddp.stats['healed'] >= 2 * len(list(net.parameters())) and ddp.stats['copied_in'] == 0
    # a gradient produced while p.grad was detached AFTER forward is copied into its slot by the hook
    y = ddp(x).sum()
    for p in net.parameters():
        p.grad = None
    y.backward()
    assert ddp.stats['copied_in'] == len(list(net.parameters()))
    for a, b in zip(net.parameters(), ref.parameters()):
        torch.testing.assert_close(a.grad, b.grad)
    assert net[0].weight.grad.is_contiguous(memory_format=torch.channels_last)


def test_ddp_counts_in_place_gradients_once():
    """A kernel that writes a gradient straight into its bucket slot reports it by hand and hands autograd None; torch
    still fires the leaf's post-accumulate hook.  Counting both would launch a bucket's exchange before its other
    gradients exist (round 1 did)."""
    from yolov4_amd.ddp import BucketedDDP

    class InPlaceMul(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, holder):
            ctx.save_for_backward(x, w)
            ctx.holder = holder
            return x * w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            p = ctx.holder['p']
            launched_before = [b.launched for b in ctx.holder['ddp'].buckets]
            ctx.holder['log'].append(launched_before)
            if getattr(p, '_y4_grad_fresh', False):
                p._y4_grad_fresh = False
                p.grad.copy_((g * x).sum(0))
                p._y4_grad_ready()
                return g * w, None, None
            return g * w, (g * x).sum(0), None

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w1 = torch.nn.Parameter(torch.ones(4))
            self.w2 = torch.nn.Parameter(torch.full((4,), 2.0))
            self.box = {}

        def forward(self, x):
            h = InPlaceMul.apply(x, self.w1, dict(self.box, p=self.w1))
            return InPlaceMul.apply(h, self.w2, dict(self.box, p=self.w2))

    net = Net().train()
    ddp = BucketedDDP(net, bucket_mb=1.0)            # both parameters in ONE bucket
    assert len(ddp.buckets) == 1
    net.box.update(ddp=ddp, log=[])
    x = torch.randn(3, 4)
    for step in range(2):
        ddp.zero_grad()
        ddp(x).sum().backward()
        assert ddp.buckets[0].pending == 0 and ddp.buckets[0].launched
        # when the SECOND gradient of the bucket was being produced the bucket had not been launched yet
        assert net.box['log'][-1] == [False]
    torch.testing.assert_close(net.w2.grad, x.sum(0))
    torch.testing.assert_close(net.w1.grad, 2 * x.sum(0))


def test_fused_optimizer_zero_grad_keeps_ddp_slots():
    from yolov4_amd.ddp import BucketedDDP
    from yolov4_amd.yolo.optim.optimizers.build import FusedAdam, FusedSGD
    net = _net()
    ddp = BucketedDDP(net, bucket_mb=0.001)
    for cls in (FusedAdam, FusedSGD):
        opt = cls(net.parameters(), lr=1e-3)
        ddp(torch.randn(2, 3, 6, 6)).sum().backward()
        assert any(float(b.flat.abs().max()) > 0 for b in ddp.buckets)
        opt.zero_grad()                              # reference loop, build.py:53,69: in place here
        for b in ddp.buckets:
            assert float(b.flat.abs().max()) == 0.0
            for p, v in zip(b.params, b.views):
                assert p.grad is not None and p.grad.data_ptr() == v.data_ptr() and p._y4_grad_fresh
        with pytest.raises(Exception):
            opt.step()                               # CPU parameters: no fallback path
    lone = torch.nn.Parameter(torch.ones(3))
    lone.grad = torch.ones(3)
    FusedAdam([lone]).zero_grad()
    assert lone.grad is not None and float(lone.grad.abs().max()) == 0.0
    o = FusedAdam([lone]); o.zero_grad(set_to_none=True)
    assert lone.grad is None


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from yolov4_amd.ddp import BucketedDDP
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    net = _net()
    if rank != 0:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    ddp = BucketedDDP(net, bucket_mb=0.001, delay_allreduce=True)        # the apex call signature
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)       # wrapped FIRST, optimizer built after
    opt.zero_grad()                                                      # as build_optimizer does: grads -> None
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 3, 6, 6, generator=g)
    y = torch.randn(8, 7, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for step in range(3):                                                # the reference's loop, nothing added
        loss = ((ddp(xs) - ys) ** 2).sum()
        loss.backward()
        if step == 0 and rank == 0:
            torch.save([p.grad.clone().contiguous() for p in net.parameters()], out + '.g')
        opt.step()
        opt.zero_grad()
    if rank == 0:
        torch.save([p.detach().clone().contiguous() for p in net.parameters()], out + '.p')
    # both ranks must hold identical parameters after 3 exchanged steps
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reference_loop_with_optimizer(tmp_path):
    """world_size 2, gloo: wrap -> build optimizer -> optimizer.zero_grad() every step (with the old
    wrapper the buckets then exchanged zeros and the ranks diverged silently).  Gradients and 3 SGD-momentum steps
    must equal the single-process full-batch run with gradient / 2."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'r')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    net = _net()
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 3, 6, 6, generator=g)
    y = torch.randn(8, 7, generator=g)
    grads = None
    for step in range(3):
        opt.zero_grad()
        (((net(x) - y) ** 2).sum() / 2).backward()           # sum over both shards / world_size (apex semantics)
        if step == 0:
            grads = [p.grad.clone() for p in net.parameters()]
        opt.step()
    for a, b in zip(torch.load(out + '.g', weights_only=True), grads):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    for a, p in zip(torch.load(out + '.p', weights_only=True), net.parameters()):
        torch.testing.assert_close(a, p.detach(), rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ harness
def test_train_step_sequence_on_cpu_toy():
    """train_step = build.py:55-69 for one micro-batch: warm-up LR, loss / ACCUMULATION_STEPS, step + zero_grad every k."""
    from yolov4_amd.yolo.engine.build import _img_infos, train_step
    cfg = {'TRAIN': {'ACCUMULATION_STEPS': 4}, 'OPTIMIZER': {'LR': '1e-2'},
           'LR_SCHEDULER': {'IS_WARMUP': True, 'WARMUP_EPOCH': 2}}
    lin = torch.nn.Linear(3, 1)
    opt = torch.optim.SGD(lin.parameters(), lr=1e-2)
    steps = []
    orig = opt.step
    opt.step = lambda *a, **k: (steps.append(1), orig(*a, **k))[1]
    x = torch.ones(2, 3)
    crit = lambda out, tgt: (out ** 2).sum()
    seen = []
    for i in range(6):
        loss = train_step(cfg, lin, crit, opt, x, None, step_index=i, len_epoch=6, epoch=0)
        seen.append(opt.param_groups[0]['lr'])
    assert len(steps) == 2                                   # i = 3 (4th micro-batch) and i = 5 (last of the epoch)
    assert abs(seen[0] - 1e-2 * 1 / 12) < 1e-12 and abs(seen[5] - 1e-2 * 6 / 12) < 1e-12
    assert loss.requires_grad and all(p.grad is None or float(p.grad.abs().max()) == 0 for p in lin.parameters())
    assert _img_infos([torch.tensor([480.]), torch.tensor([640.]), torch.tensor([608.]), torch.tensor([608.]),
                       torch.tensor([17.]), torch.tensor([0.])], 1) == [[480.0, 640.0, 608.0, 608.0, 17.0, 0.0]]


def test_build_optimizer_sgd_branch_and_no_detached_grads():
    from yolov4_amd.yolo.model.yolov4 import YOLOv4
    from yolov4_amd.yolo.optim.optimizers.build import FusedSGD, build_optimizer
    cfg = {'OPTIMIZER': {'TYPE': 'SGD', 'LR': '1e-3', 'MOMENTUM': 0.9, 'DECAY': 1e-5, 'NO_BIAS': True, 'NO_NORM': True}}
    m = YOLOv4(recipe.MODEL_CFG)
    opt = build_optimizer(cfg, m)
    assert isinstance(opt, FusedSGD) and opt.defaults['momentum'] == 0.9 and opt.defaults['weight_decay'] == 1e-5
    assert opt.param_groups[1]['weight_decay'] == 0.


def test_roctx_ranges_are_noops_unless_enabled(monkeypatch):
    """yolov4_amd.trace: with Y4_ROCTX unset the ranges do nothing and no library is loaded; with Y4_ROCTX=1 they call
    roctxRangePushA / roctxRangePop of libroctx64.so (present in the ROCm image) in matched pairs."""
    from yolov4_amd import trace
    monkeypatch.delenv('Y4_ROCTX', raising=False)
    trace._STATE.update(lib=None, tried=False)
    with trace.range('x'):
        pass
    assert trace._STATE['tried'] is False
    monkeypatch.setenv('Y4_ROCTX', '1')
    calls = []

    class Fake:
        def roctxRangePushA(self, name):
            calls.append(('push', name))
            return 0

        def roctxRangePop(self):
            calls.append(('pop', None))
            return 0
    trace._STATE.update(lib=Fake(), tried=True)
    try:
        with trace.range('y4.forward'):
            with trace.range('inner'):
                pass
    finally:
        trace._STATE.update(lib=None, tried=False)
    assert calls == [('push', b'y4.forward'), ('push', b'inner'), ('pop', None), ('pop', None)]


def test_bench_launches_its_own_ranks_when_not_under_torchrun():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (VERDICT r3 missing #1): the parent starts the two
    ranks as child processes (torch.distributed.run, 127.0.0.1), relays rank 0's ONE JSON line on stdout and returns the
    workers' status.  --stub-step swaps the GPU step for a gloo all-reduce so that the plumbing runs here: the mean of the
    per-rank 'gradients' (1, 2) is 1.5 only if both ranks met in the collective."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--stub-step'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode().strip().splitlines()
    assert len(out) == 1, out                                    # stdout carries the JSON line and nothing else
    line = json.loads(out[0])
    assert line['stub'] is True and line['n_gpus'] == 2 and line['steps'] == 3 and abs(line['mean_grad'] - 1.5) < 1e-6
    # a failing worker fails the command: without a GPU the real step dies in every rank, and the parent says so
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    if not torch.cuda.is_available():
        assert r.returncode != 0 and r.stdout.decode().strip() == ''
