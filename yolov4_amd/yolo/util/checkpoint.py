# -*- coding: utf-8 -*-
"""Checkpoint I/O compatible with the reference (yolo/util/utils.py:17-24, main_amp.py:140-168,
220-229, val.py:78-83): `{'epoch','ap50','ap50_95','best_ap50','best_ap50_95','state_dict',
'optimizer','lr_scheduler'}` dictionaries, optional `module.` prefix on the state_dict keys, OIHW conv
weights on the wire (KRSC only in device memory)."""
import os
import shutil
from collections import OrderedDict

import torch


def save_checkpoint(state, is_best, filename='checkpoint.pth.tar', output_dir='./'):
    os.makedirs(output_dir, exist_ok=True)
    wire = dict(state)
    if 'state_dict' in wire:        # plain contiguous tensors, as a reference checkpoint holds them
        wire['state_dict'] = OrderedDict((k, v.detach().cpu().contiguous()) for k, v in wire['state_dict'].items())
    path = os.path.join(output_dir, filename)
    torch.save(wire, path)
    if is_best:
        shutil.copyfile(path, os.path.join(output_dir, 'model_best.pth.tar'))
    return path


def strip_module_prefix(state_dict):
    return OrderedDict((k[len('module.'):] if k.startswith('module.') else k, v) for k, v in state_dict.items())


def load_checkpoint(model, path, strict=True):
    """Loads a reference-format checkpoint (only with weights_only=True: nothing in the file is executed)
    and returns the checkpoint dict."""
    ckpt = torch.load(path, map_location='cpu', weights_only=True)
    model.load_state_dict(strip_module_prefix(ckpt['state_dict']), strict=strict)
    return ckpt
