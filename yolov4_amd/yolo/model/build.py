# -*- coding: utf-8 -*-
"""Factories with the reference's signatures (yolo/model/build.py:19-33)."""
from argparse import Namespace
from typing import Dict

from .yolov4 import YOLOv4
from .yololoss import YOLOLoss


def build_model(args: Namespace, cfg: Dict, device=None):
    # activations and filters are NHWC / KRSC on this path whatever args.channels_last says
    # (the flag only selected a memory format in the reference, build.py:20-26)
    model = YOLOv4(cfg['MODEL'], device=device)
    return model.to(device=device)


def build_criterion(cfg: Dict, device=None):
    return YOLOLoss(cfg['MODEL'], ignore_thresh=float(cfg['CRITERION']['IGNORE_THRESH']), device=device).to(device)
