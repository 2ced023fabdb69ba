# -*- coding: utf-8 -*-
"""Train / validate harness with the reference's call sequence (yolo/engine/build.py:41-107, :111-190), built from
the hot-path pieces of this package.  What is NOT here, on purpose: apex amp (`amp.scale_loss`, the reference
trains at opt-level O0 = plain fp32), tqdm, and COCOeval (pycocotools; pass an `evaluator` callable).

    train(args, cfg, train_loader, model, criterion, optimizer, device, epoch)
    validate(val_loader, model, conf_threshold, nms_threshold, device)

`train_step` is the body of the reference's loop for one micro-batch; `train` loops it.  Under BucketedDDP the
gradient exchange is skipped on all but the last micro-step of an accumulation window (apex all-reduces after every
backward; the result is the same).
"""
import time

import torch
import torch.distributed as dist

from ..optim.lr_schedulers.build import adjust_learning_rate
from ... import trace
from ..util.utils import detections_to_coco, postprocess


class AverageMeter:
    """yolo/util/metric.py:11-27"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def reduce_tensor(args, tensor):
    """build.py:193-197: sum over ranks / world_size (logging only)."""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= getattr(args, 'world_size', dist.get_world_size())
    return rt


def train_step(cfg, model, criterion, optimizer, input, target, device=None, step_index=0, len_epoch=None, epoch=0):
    """One micro-batch of build.py:55-69.  Returns the (already divided) loss tensor.

    step_index: i of the reference loop; the optimizer steps when (i + 1) % ACCUMULATION_STEPS == 0 or at the last
    batch of the epoch (len_epoch)."""
    accumulation_steps = int(cfg['TRAIN']['ACCUMULATION_STEPS'])
    if cfg['LR_SCHEDULER']['IS_WARMUP'] and epoch < int(cfg['LR_SCHEDULER']['WARMUP_EPOCH']) and len_epoch:
        adjust_learning_rate(cfg, optimizer, epoch, step_index, len_epoch)                       # :56-57
    steps_now = (step_index + 1) % accumulation_steps == 0 or (len_epoch is not None and step_index + 1 == len_epoch)
    if hasattr(model, 'accumulating'):
        model.accumulating = not steps_now              # BucketedDDP: exchange once per window, on its last backward
    if device is not None:
        input = input.to(device)
    with trace.range('y4.forward'):
        output = model(input)                                                                    # :60
    with trace.range('y4.loss'):
        loss = criterion(output, target) / accumulation_steps                                   # :61
    with trace.range('y4.backward'):
        loss.backward()                                                                          # :64-65 (O0: scale 1)
    if steps_now:
        with trace.range('y4.optimizer'):
            optimizer.step()                                                                     # :67-69
            optimizer.zero_grad()
    return loss


def train(args, cfg, train_loader, model, criterion, optimizer, device=None, epoch=0, log=None):
    """build.py:41-107.  `log`: optional callable(str) (the reference logs through its rank-0 logger)."""
    batch_time, losses = AverageMeter(), AverageMeter()
    model.train()
    end = time.time()
    optimizer.zero_grad()
    n = len(train_loader)
    print_freq = int(getattr(args, 'print_freq', 10))
    world = int(getattr(args, 'world_size', 1))
    for i, (input, target) in enumerate(train_loader):
        loss = train_step(cfg, model, criterion, optimizer, input, target, device, i, n, epoch)
        if (i + 1) % print_freq == 0:
            reduced = reduce_tensor(args, loss.data) if getattr(args, 'distributed', False) else loss.data
            losses.update(float(reduced), input.size(0))                   # host <-> device sync, as the reference
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            batch_time.update((time.time() - end) / print_freq)
            end = time.time()
            if log is not None:
                bs = float(cfg['DATA']['BATCH_SIZE'])
                log(f'Epoch: [{epoch + 1}][{i + 1}/{n}]\tTime {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                    f'Speed {world * bs / batch_time.val:.3f} ({world * bs / batch_time.avg:.3f})\t'
                    f"Lr {optimizer.param_groups[0]['lr']:.8f}\tLoss {losses.val:.10f} ({losses.avg:.4f})")
    return losses.avg


def _img_infos(info, batch):
    """target['img_info'] as the reference's loader collates it (a list of per-field tensors of shape [B], :124) or as
    a [B, n] / [n] tensor or nested list -> list of B rows of floats."""
    if torch.is_tensor(info):
        rows = info.reshape(1, -1) if info.dim() == 1 else info
        return [[float(v) for v in r] for r in rows.cpu()]
    if len(info) and torch.is_tensor(info[0]) and info[0].dim() >= 1 and info[0].numel() == batch and len(info) != batch:
        cols = [c.reshape(-1).cpu() for c in info]                      # default collate: one tensor per field
        return [[float(c[b]) for c in cols] for b in range(batch)]
    if len(info) and isinstance(info[0], (list, tuple)) or (torch.is_tensor(info[0]) and info[0].dim() >= 1 and info[0].numel() > 1):
        return [[float(v) for v in r] for r in info]
    return [[float(v) for v in info]]


@torch.no_grad()
def validate(val_loader, model, conf_threshold, nms_threshold, device=None, evaluator=None, num_classes=80):
    """build.py:111-190.  Returns (AP50_95, AP50) from `evaluator(records, image_ids)` when one is given (the
    reference calls pycocotools' COCOeval there); without one returns (0, 0) like the reference's empty branch and
    leaves the COCO-format records on `validate.records`.  Unlike the reference the loader may hand batches larger
    than 1: target['img_info'] is then a list of per-image [h, w, S, S, id, ...] rows."""
    model.eval()
    ids, data_list = [], []
    class_ids = getattr(getattr(val_loader, 'dataset', None), 'class_ids', None)
    for img, target in val_loader:
        assert isinstance(target, dict)
        infos = _img_infos(target['img_info'], img.shape[0])
        with trace.range('y4.eval_forward'):
            outputs = model(img.to(device) if device is not None else img)                        # :133
        with trace.range('y4.postprocess'):
            outputs = postprocess(outputs, num_classes, conf_threshold, nms_threshold)            # :137
        for det, info in zip(outputs, infos):
            id_ = int(info[-2])
            ids.append(id_)
            kw = {'class_ids': class_ids} if class_ids is not None else {}
            data_list.extend(detections_to_coco(det, info[:4], id_, **kw))                        # :144-164
    validate.records = data_list
    if evaluator is not None and data_list:
        return evaluator(data_list, ids)
    return 0, 0
