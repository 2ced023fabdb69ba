# -*- coding: utf-8 -*-
"""yolov4_amd -- MI355X (gfx950) native forward/backward path for YOLOv4.

Drop-in for the hot-path modules of zjykzj/YOLOv4:

    reference import                              this package
    darknet.darknet.ConvBNAct / ResBlock / ...    yolov4_amd.darknet.darknet
    yolo.model.yolov4.YOLOv4                      yolov4_amd.yolo.model.yolov4
    yolo.model.yololayer.YOLOLayer                yolov4_amd.yolo.model.yololayer
    yolo.model.yololoss.YOLOLoss                  yolov4_amd.yolo.model.yololoss
    yolo.model.build.build_model/build_criterion  yolov4_amd.yolo.model.build
    yolo.util.utils.postprocess / nms             yolov4_amd.yolo.util.utils

All arithmetic runs in libyolov4_amd.so (hand-written HIP, C ABI in
include/yolov4_amd.h); there is no CPU or PyTorch-op fallback.
"""
from ._lib import LIB_PATH, Y4Error, lib  # noqa: F401

__version__ = '0.1.0'


def set_conv_mode(mode):
    """0 / 'f32': fp32 MFMA (exact fma chain).  1 / 'bf16x3': exact 3-way bf16 split, six bf16 MFMAs per product.
    3 / 'f16x2' (default): two fp16 pieces per operand, three MFMAs per product (all three fp32-grade, see
    include/yolov4_amd.h).  'bf16' (BASELINE configs[4], mixed precision): mode 3 with bf16 operands on the plane layers --
    one bf16 MFMA per product where the flops are, the fp32-grade HBM-bound kernels elsewhere (y4_set_planes_bf16).
    2 / 'bf16_all': EVERY conv operand rounded to bf16 (the plane layers on the same bf16 DMA kernels)."""
    from ._lib import check
    hybrid = mode == 'bf16'
    m = {'f32': 0, 'bf16x3': 1, 'bf16_all': 2, 'bf16': 3, 'f16x2': 3}.get(mode, mode)
    check(lib().y4_set_conv_mode(int(m)), 'set_conv_mode')
    check(lib().y4_set_planes_bf16(1 if hybrid else 0), 'set_planes_bf16')


def get_conv_mode():
    return lib().y4_get_conv_mode()
