# -*- coding: utf-8 -*-
"""Optimizer factory with the reference's signature (yolo/optim/optimizers/build.py:18-80):
same two parameter groups (decay / no-decay by `filter_weight`), ADAM backed by the fused HIP step."""
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


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad), one fused HIP sweep per parameter block."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = lib()
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise ops.Y4Error('FusedAdam: parameters must live on the GPU (no CPU fallback)')
                g = p.grad
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if g.stride() != p.stride():                      # same dense memory order as the parameter
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                st['step'] += 1
                check(L.y4_adam_step_f32(ops._ptr(p), ops._ptr(g), ops._ptr(st['exp_avg']), ops._ptr(st['exp_avg_sq']),
                                         p.numel(), float(group['lr']), float(b1), float(b2), float(group['eps']),
                                         float(group['weight_decay']), int(st['step']), float(self.grad_scale),
                                         ops._stream()), 'adam_step')
        return loss


def build_optimizer(cfg: Dict, model: Module):
    optimizer_type = cfg['OPTIMIZER']['TYPE']
    lr = float(cfg['OPTIMIZER']['LR'])
    groups = filter_weight(cfg, model)
    if 'ADAM' == optimizer_type:
        optimizer = FusedAdam(groups, lr=lr, betas=(0.9, 0.999), eps=1e-08)      # adam.py:14-15
    else:
        raise ValueError(f"{optimizer_type} does not support.")              # SGD: not on this path yet
    optimizer.zero_grad()
    return optimizer
