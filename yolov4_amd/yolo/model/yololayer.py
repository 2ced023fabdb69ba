# -*- coding: utf-8 -*-
"""YOLO head decode behind the reference's YOLOLayer API
(yolo/model/yololayer.py:16-166), one fused HIP kernel per direction."""
import numpy as np
import torch
from torch import nn

from ... import ops


class YOLOLayer(nn.Module):
    strides = [8, 16, 32]

    def __init__(self, cfg, layer_no, device=None):
        super().__init__()
        self.stride = self.strides[layer_no]
        self.layer_no = layer_no
        self.anchors = cfg['ANCHORS']
        self.anchor_mask = cfg['ANCHOR_MASK'][layer_no]
        self.n_anchors = len(self.anchor_mask)
        self.all_anchors_grid = [(w / self.stride, h / self.stride) for w, h in self.anchors]
        self.masked_anchors = torch.from_numpy(np.array([self.all_anchors_grid[i] for i in self.anchor_mask]))
        self.n_classes = cfg['N_CLASSES']
        self.device = device

    def _anchors_f32(self):
        # float64 grid-unit anchors cast to the activation dtype at use (yololayer.py:117-120)
        return [(float(np.float32(w)), float(np.float32(h))) for w, h in self.masked_anchors.tolist()]

    def forward(self, output):
        if self.training:
            out, pred = ops.YoloDecodeTrainFn.apply(output, self._anchors_f32(), self.n_classes)
            return {'layer_no': self.layer_no, 'output': out, 'pred': pred}
        return ops.yolo_decode_eval(output, self._anchors_f32(), self.n_classes, self.stride)

    def decode_into(self, logits, out, n_total, box_off):
        """eval decode straight into rows [box_off, ...) of the concatenated [B, n_total, 5+C] buffer"""
        return ops.yolo_decode_eval(logits, self._anchors_f32(), self.n_classes, self.stride, out, n_total, box_off)
