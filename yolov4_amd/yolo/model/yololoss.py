# -*- coding: utf-8 -*-
"""Detection loss behind the reference's YOLOLoss API (yolo/model/yololoss.py:94-443):
YOLOv3-style weighted-BCE(xy) + MSE(wh)/2 + BCE(obj) + BCE(cls), summed over batch and
the 3 layers (no CIoU / focal anywhere in the reference)."""
from typing import Dict

import numpy as np
import torch
from torch import nn

from ... import ops


def bboxes_iou(bboxes_a, bboxes_b, xyxy=True):
    """Pairwise IoU [Na,4] x [Nb,4] -> [Na,Nb] (yololoss.py:16-91) as one HIP kernel (y4_bboxes_iou_f32).  The loss
    path does not call it: target assignment and the ignore mask carry the same arithmetic inlined."""
    return ops.bboxes_iou_raw(bboxes_a, bboxes_b, xyxy)


class YOLOLoss(nn.Module):
    strides = [8, 16, 32]

    def __init__(self, cfg: Dict, ignore_thresh=0.7, device=None, mutate_outputs=True):
        super().__init__()
        self.cfg = cfg
        self.ignore_thresh = ignore_thresh
        self.device = device
        self.anchors = cfg['ANCHORS']
        self.n_classes = cfg['N_CLASSES']
        # yololoss.py:402-407 multiplies the masks into outputs[*]['output'] in place; kept by default
        self.mutate_outputs = mutate_outputs
        self.last = [None, None, None]

    def _layer_cfg(self, layer_no):
        st = self.strides[layer_no]
        # anchors / stride in float64, used as fp32 (yololoss.py:155-167)
        grid = [(float(np.float32(w / st)), float(np.float32(h / st))) for w, h in self.anchors]
        return {'stride': st, 'ignore_thresh': float(np.float32(self.ignore_thresh)), 'all_anchors': grid,
                'anch_mask': list(self.cfg['ANCHOR_MASK'][layer_no]), 'mutate_output': self.mutate_outputs}

    def build_target(self, output, pred, layer_no, labels):
        """Dense (target, obj_mask, tgt_mask, tgt_scale) as yololoss.py:118-371 returns them."""
        cfg = self._layer_cfg(layer_no)
        cfg['mutate_output'] = False
        ops.YoloLossLayerFn.apply(output.detach().contiguous(), pred, labels, cfg)
        target, tgt_mask, tgt_scale = ops.yolo_loss_dense_targets(cfg['last'])
        return target, cfg['last']['obj_mask'], tgt_mask, tgt_scale

    def forward(self, outputs, targets):
        assert isinstance(outputs, list)
        assert isinstance(targets, dict)
        labels = targets['padded_labels']
        total = None
        for od in outputs:
            assert isinstance(od, dict)
            layer_no = od['layer_no']
            cfg = self._layer_cfg(layer_no)
            loss = ops.YoloLossLayerFn.apply(od['output'], od['pred'], labels, cfg)
            self.last[layer_no] = cfg['last']
            total = loss if total is None else total + loss
        return total
