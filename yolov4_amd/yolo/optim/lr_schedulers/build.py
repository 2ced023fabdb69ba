# -*- coding: utf-8 -*-
"""LR schedule helpers with the reference's signatures (yolo/optim/lr_schedulers/build.py:17-54):
host scalar arithmetic only."""
from typing import Dict

from torch.optim.lr_scheduler import CosineAnnealingLR, MultiStepLR
from torch.optim.optimizer import Optimizer


def adjust_learning_rate(cfg: Dict, optimizer: Optimizer, epoch: int, step: int, len_epoch: int) -> None:
    """Linear warm-up over WARMUP_EPOCH epochs (build.py:17-27)."""
    lr = float(cfg['OPTIMIZER']['LR'])
    warmup_epoch = int(cfg['LR_SCHEDULER']['WARMUP_EPOCH'])
    if epoch < warmup_epoch:
        lr = lr * float(1 + step + epoch * len_epoch) / (warmup_epoch * len_epoch)
    for group in optimizer.param_groups:
        group['lr'] = lr


def build_lr_scheduler(cfg: Dict, optimizer: Optimizer):
    assert isinstance(optimizer, Optimizer)
    kind = cfg['LR_SCHEDULER']['TYPE']
    warm = int(cfg['LR_SCHEDULER']['WARMUP_EPOCH']) if cfg['LR_SCHEDULER']['IS_WARMUP'] else 0
    if kind == 'MultiStepLR':
        milestones = [int(m) - warm for m in cfg['LR_SCHEDULER']['MILESTONES']]
        return MultiStepLR(optimizer, milestones=milestones, gamma=float(cfg['LR_SCHEDULER']['GAMMA']))
    if kind == 'CosineAnnealingLR':
        return CosineAnnealingLR(optimizer, T_max=int(cfg['TRAIN']['MAX_EPOCHS']) - warm,
                                 eta_min=float(cfg['LR_SCHEDULER']['MINIMAL_LR']))
    raise ValueError(f"{kind} does not support.")
