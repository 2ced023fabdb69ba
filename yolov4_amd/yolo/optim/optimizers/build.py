# -*- coding: utf-8 -*-
"""Optimizer factory with the reference's signature (yolo/optim/optimizers/build.py:18-80):
same two parameter groups (decay / no-decay by `filter_weight`), ADAM backed by the fused HIP step."""
import ctypes
from typing import Dict

import torch
from torch import nn
from torch.nn import Module

from .... import ops
from ...._lib import check, lib


def filter_weight(cfg: Dict, module: Module):
    """build.py:38-80: conv/linear weights decay; biases (NO_BIAS) and norm params (NO_NORM) do not."""
    decay, no_decay = [], []
    no_bias = cfg['OPTIMIZER']['NO_BIAS'] is True
    no_norm = cfg['OPTIMIZER']['NO_NORM'] is True
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.modules.conv._ConvNd)):
            decay.append(m.weight)
            if m.bias is not None:
                (no_decay if no_bias else decay).append(m.bias)
        elif isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.GroupNorm, nn.LayerNorm)):
            for p in (m.weight, m.bias):
                if p is not None:
                    (no_decay if no_norm else decay).append(p)
    assert len(list(module.parameters())) == len(decay) + len(no_decay)
    return [dict(params=[p for p in decay if p.requires_grad]),
            dict(params=[p for p in no_decay if p.requires_grad], weight_decay=0.)]


CHUNK = 1 << 16          # elements per table record of the multi-tensor kernels


class _MultiTensorOptimizer(torch.optim.Optimizer):
    """Shared plumbing of the fused optimizers: every parameter, its gradient and its state tensors are cut into
    runs of CHUNK elements listed in ONE device table ({p, g, m, v, n, hyper row} records, include/yolov4_amd.h);
    a step is one kernel launch over that table.  The table is rebuilt only when a pointer changes (gradients that
    live in BucketedDDP's flat buckets never move, so in steady state nothing is uploaded but 16 floats of
    per-group scalars)."""

    n_state = 2

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self.grad_scale = 1.0
        self._table = None
        self._table_key = None
        self.launches = 0            # kernel launches issued by step() so far (tests / profiling)

    def zero_grad(self, set_to_none: bool = False):
        """Default differs from torch (set_to_none=True): gradients that live in BucketedDDP's flat buckets are
        cleared in place (one memset per bucket) so that the slots stay attached; everything else is zeroed as
        torch does."""
        done = set()
        rest = []
        for group in self.param_groups:
            for p in group['params']:
                ddp = getattr(p, '_y4_ddp', None)
                if ddp is not None and not set_to_none:
                    if id(ddp) not in done:
                        done.add(id(ddp))
                        ddp.zero_grad()
                elif p.grad is not None:
                    rest.append(p)
        for p in rest:
            if set_to_none:
                p.grad = None
            else:
                p.grad.detach_()
                p.grad.requires_grad_(False)
                p.grad.zero_()

    @staticmethod
    def _same_layout(t, p):
        return t.stride() == p.stride() and t.dtype == p.dtype and t.device == p.device

    def _dense_like(self, t, p):
        """t re-laid in p's dense memory order (the kernels pair elements by raw offset): e.g. Adam moments
        restored by load_state_dict from a reference checkpoint are contiguous OIHW while the parameter is KRSC."""
        if self._same_layout(t, p):
            return t
        return torch.empty_like(p, memory_format=torch.preserve_format).copy_(t)

    def _state_tensors(self, p, group):
        raise NotImplementedError

    def _hyper_row(self, group, step):
        raise NotImplementedError

    def _launch(self, table, nchunks, hyper, nh, group0):
        raise NotImplementedError

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        rows, key, hyper, hyper_idx, keep, dev = [], [], [], {}, [], None
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise ops.Y4Error(f'{type(self).__name__}: parameters must live on the GPU (no CPU fallback)')
                if p.dtype != torch.float32:
                    raise ops.Y4Error(f'{type(self).__name__}: float32 parameters only')
                if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                    raise ops.Y4Error(f'{type(self).__name__}: parameters must be dense')
                g = self._dense_like(p.grad, p)
                states, step = self._state_tensors(p, group)
                hk = (gi, step) + tuple(float(group[k]) for k in self._hyper_keys)
                if hk not in hyper_idx:
                    hyper_idx[hk] = len(hyper)
                    hyper.append(self._hyper_row(group, step))
                h = hyper_idx[hk]
                ptrs = [p.data_ptr(), g.data_ptr()] + [t.data_ptr() for t in states] + [0] * (2 - len(states))
                n = p.numel()
                key.append((ptrs[0], ptrs[1], ptrs[2], ptrs[3], n, h))
                for o in range(0, n, CHUNK):
                    rows.append((ptrs[0] + 4 * o, ptrs[1] + 4 * o, ptrs[2] + 4 * o, (ptrs[3] + 4 * o) if ptrs[3] else 0,
                                 min(CHUNK, n - o), h))
                keep.append(g)                               # a repacked gradient stays alive until the launch is enqueued
                dev = p.device
        if not rows:
            return loss
        if len(hyper) > 16:
            raise ops.Y4Error('more than 16 distinct (group, step) combinations in one optimizer step')
        key = tuple(key)
        if key != self._table_key:
            import numpy as np
            host = torch.from_numpy(np.asarray(rows, dtype=np.int64))
            self._table = host.to(dev)
            self._table_key = key
        flat = [v for row in hyper for v in row]
        self._launch(self._table, len(rows), ops.float_array(flat), len(hyper))
        self.launches += 1
        del keep
        return loss


class FusedAdam(_MultiTensorOptimizer):
    """torch.optim.Adam semantics (no amsgrad): ONE multi-tensor HIP launch per step (y4_adam_multi_step_f32)."""

    _hyper_keys = ('lr', 'weight_decay')

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _state_tensors(self, p, group):
        st = self.state[p]
        if not st:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        for k in ('exp_avg', 'exp_avg_sq'):
            if not self._same_layout(st[k], p):
                st[k] = self._dense_like(st[k].to(device=p.device, dtype=p.dtype), p)
        step = int(st['step']) + 1              # torch.optim.Adam keeps `step` as a tensor; a restored state has one
        st['step'] = step
        return [st['exp_avg'], st['exp_avg_sq']], step

    def _hyper_row(self, group, step):
        b1, b2 = group['betas']
        out = (ctypes.c_float * 4)()
        check(lib().y4_adam_hyper_f32(float(group['lr']), float(b1), float(b2), float(group['weight_decay']), step, out),
              'adam_hyper')
        return list(out)

    def _launch(self, table, nchunks, hyper, nh):
        g0 = self.param_groups[0]
        if any(g['betas'] != g0['betas'] or g['eps'] != g0['eps'] for g in self.param_groups):
            raise ops.Y4Error('FusedAdam: betas / eps must be the same in every param group')
        b1, b2 = g0['betas']
        check(lib().y4_adam_multi_step_f32(ops._ptr(table), nchunks, hyper, nh, float(b1), float(b2), float(g0['eps']),
                                           float(self.grad_scale), ops._stream()), 'adam_multi_step')


class FusedSGD(_MultiTensorOptimizer):
    """torch.optim.SGD(lr, momentum, weight_decay) semantics (dampening 0, no nesterov), one launch per step."""

    _hyper_keys = ('lr', 'momentum', 'weight_decay')

    def __init__(self, params, lr=0.1, momentum=0.9, weight_decay=1e-5):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    def _state_tensors(self, p, group):
        st = self.state[p]
        first = 'momentum_buffer' not in st or st['momentum_buffer'] is None
        if first:
            st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not self._same_layout(st['momentum_buffer'], p):
            st['momentum_buffer'] = self._dense_like(st['momentum_buffer'].to(device=p.device, dtype=p.dtype), p)
        return [st['momentum_buffer']], (1 if first else 2)

    def _hyper_row(self, group, step):
        return [float(group['lr']), float(group['momentum']), float(group['weight_decay']), 1.0 if step == 1 else 0.0]

    def _launch(self, table, nchunks, hyper, nh):
        check(lib().y4_sgd_multi_step_f32(ops._ptr(table), nchunks, hyper, nh, float(self.grad_scale), ops._stream()),
              'sgd_multi_step')


def build_optimizer(cfg: Dict, model: Module):
    optimizer_type = cfg['OPTIMIZER']['TYPE']
    lr = float(cfg['OPTIMIZER']['LR'])
    groups = filter_weight(cfg, model)
    if 'SGD' == optimizer_type:                                                # build.py:25-28, sgd.py:14-15
        optimizer = FusedSGD(groups, lr=lr, momentum=float(cfg['OPTIMIZER']['MOMENTUM']),
                             weight_decay=float(cfg['OPTIMIZER']['DECAY']))
    elif 'ADAM' == optimizer_type:
        optimizer = FusedAdam(groups, lr=lr, betas=(0.9, 0.999), eps=1e-08)      # adam.py:14-15
    else:
        raise ValueError(f"{optimizer_type} does not support.")
    optimizer.zero_grad()            # build.py:34; in place: gradient slots of a BucketedDDP wrapper stay attached
    return optimizer
