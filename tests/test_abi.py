# -*- coding: utf-8 -*-
"""CPU-side checks of the C-ABI boundary and the host mirror of the reference
interface: the library loads without a GPU, exports every symbol the header
declares, and the Python classes keep the reference's constructor signatures,
attribute tree and state_dict -- and fail loudly instead of falling back."""
import inspect
import os
import re

import numpy as np
import pytest
import torch

import recipe
import yolov4_amd
from yolov4_amd import _lib, ops
from yolov4_amd.darknet.darknet import ConvBNAct, CSPDownSample, CSPDownSample0, ResBlock
from yolov4_amd.yolo.model.build import build_criterion, build_model
from yolov4_amd.yolo.model.yololayer import YOLOLayer
from yolov4_amd.yolo.model.yololoss import YOLOLoss
from yolov4_amd.yolo.model.yolov4 import YOLOv4
from yolov4_amd.yolo.util import utils
from oracle import network as NW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'yolov4_amd.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(y4_\w+)\s*\(', txt)))


def test_library_loads_and_exports_every_declared_symbol():
    L = yolov4_amd.lib()
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), f'{s} declared in include/yolov4_amd.h but not exported'
    assert sorted(_lib.PROTOTYPES) == syms, 'ctypes prototypes and header disagree'
    assert L.y4_version() >= 100
    assert L.y4_strerror(0) == b'ok' and b'shape' in L.y4_strerror(1)
    assert L.y4_device_count() >= 0
    assert L.y4_get_conv_mode() in (0, 1, 2, 3) and L.y4_set_conv_mode(7) == 1


def test_workspace_queries_run_on_host():
    L = yolov4_amd.lib()
    assert L.y4_conv2d_dgrad_workspace(128, 255, 3) == 128 * 9 * 256 * 6 + 64 + 4096
    assert L.y4_conv2d_fwd_workspace(128, 255, 3) == 64 + 4096 + 255 * 9 * 128 * 6
    assert L.y4_bn_workspace(1000, 64) >= 2 * 64 * 8
    assert L.y4_conv2d_wgrad_workspace(64, 76, 76, 128, 128, 3, 1) > 0
    assert L.y4_yolo_loss_workspace(4, 76, 3, 60, 80) > 4 * 3 * 76 * 76 * 4
    assert L.y4_post_nms_workspace(1000, 160) >= 1000 * (8 + 16 + 4 + 4)


def test_plane_window_query_runs_on_host():
    """y4_conv_planes_fit: the DMA kernels' 32-bit buffer windows, asked by the host before a producer writes planes."""
    L = yolov4_amd.lib()
    was = L.y4_get_conv_mode()
    try:
        L.y4_set_conv_mode(3)
        assert L.y4_conv_planes_fit(64, 76, 76, 128, 128, 3, 1, 1) == 1
        assert L.y4_conv_planes_fit(64, 152, 152, 128, 256, 3, 2, 1) == 1
        assert L.y4_conv_planes_fit(64, 76, 76, 100, 128, 3, 1, 1) == 0            # not whole K tiles
        assert L.y4_conv_planes_fit(64, 75, 76, 128, 256, 3, 2, 1) == 0            # stride 2 on an odd grid
        assert L.y4_conv_planes_fit(1, 4096, 4096, 64, 128, 3, 1, 1) == 0          # one image = 4 GiB: beyond a forward window
        assert L.y4_conv_planes_fit(1, 1024, 1024, 64, 128, 3, 1, 1) == 1
        # stride 2: the plane dgrad scatters into ONE window over the whole dx tensor (6 GiB here); forward and wgrad fit
        assert L.y4_conv_planes_fit(512, 152, 152, 128, 256, 3, 2, 1) == 0
        assert L.y4_conv_planes_fit(512, 152, 152, 128, 256, 3, 2, 0) == 1
        L.y4_set_conv_mode(0)
        assert L.y4_conv_planes_fit(64, 76, 76, 128, 128, 3, 1, 1) == 0            # no plane kernels in this mode
    finally:
        L.y4_set_conv_mode(was)


def test_null_and_shape_errors_are_reported_before_any_launch():
    L = yolov4_amd.lib()
    assert L.y4_conv2d_fwd_f32(None, 32, None, None, 32, 1, 8, 8, 32, 32, 3, 1, None, None, 0, None, 0, None, None, None, 0, None) == 2
    # Cin not a multiple of 32 -> shape error (pointers are fake but never dereferenced on the host)
    fake = 0x1000
    assert L.y4_conv2d_fwd_f32(fake, 48, fake, fake, 32, 1, 8, 8, 48, 32, 3, 1, None, None, 0, None, 0, None, None, fake, 1 << 30, None) == 1
    assert L.y4_conv2d_fwd_f32(fake, 32, fake, fake, 32, 1, 8, 8, 32, 32, 5, 1, None, None, 0, None, 0, None, None, fake, 1 << 30, None) == 1


def test_state_dict_matches_reference_tree():
    m = YOLOv4(recipe.MODEL_CFG)
    sd = m.state_dict()
    spec = NW.yolov4_state_dict_spec()
    assert list(sd.keys()) == [k for k, _ in spec]
    for k, shp in spec:
        assert tuple(sd[k].shape) == tuple(shp)
        assert sd[k].dtype == (torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
    assert len(sd) == 648 and sum(p.numel() for p in m.parameters()) == 64885341
    # round trip through the reference's OIHW layout keeps KRSC memory
    m2 = YOLOv4(recipe.MODEL_CFG)
    recipe.fill_state_dict_(sd, 3)
    m2.load_state_dict({k: v.contiguous() for k, v in sd.items()})
    w = m2.backbone.stage1.base.conv.weight
    assert w.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(m2.state_dict()['backbone.stage1.base.conv.weight'], sd['backbone.stage1.base.conv.weight'])


def test_constructor_signatures_match_reference():
    assert list(inspect.signature(ConvBNAct.__init__).parameters)[1:] == \
        ['in_ch', 'out_ch', 'kernel_size', 'stride', 'bias', 'bn', 'act']
    assert list(inspect.signature(ResBlock.__init__).parameters)[1:] == ['ch', 'num_blocks', 'shortcut', 'act']
    assert list(inspect.signature(CSPDownSample0.__init__).parameters)[1:] == \
        ['in_ch', 'out_ch', 'kernel_size', 'stride', 'act']
    assert list(inspect.signature(CSPDownSample.__init__).parameters)[1:] == \
        ['in_ch', 'out_ch', 'kernel_size', 'stride', 'num_blocks', 'shortcut', 'act']
    assert list(inspect.signature(YOLOv4.__init__).parameters)[1:] == ['cfg', 'device']
    assert list(inspect.signature(YOLOLayer.__init__).parameters)[1:] == ['cfg', 'layer_no', 'device']
    assert list(inspect.signature(YOLOLoss.__init__).parameters)[1:4] == ['cfg', 'ignore_thresh', 'device']
    assert list(inspect.signature(utils.postprocess).parameters) == ['prediction', 'num_classes', 'conf_thre', 'nms_thre']
    assert list(inspect.signature(utils.nms).parameters) == ['bbox', 'thresh', 'score', 'limit']
    assert YOLOLayer.strides == [8, 16, 32] and YOLOLoss.strides == [8, 16, 32]
    with pytest.raises(ValueError):
        ConvBNAct(8, 8, 1, 1, act='gelu')
    with pytest.raises(AssertionError):
        YOLOv4(dict(recipe.MODEL_CFG, TYPE='YOLOv3'))
    c = ConvBNAct(32, 255, 1, 1, bias=True, bn=False, act='linear')
    assert sorted(c.state_dict()) == ['conv.bias', 'conv.weight']
    c = ConvBNAct(32, 64, 3, 2)
    assert sorted(c.state_dict()) == ['conv.weight', 'norm.bias', 'norm.num_batches_tracked', 'norm.running_mean',
                                      'norm.running_var', 'norm.weight']


def test_no_cpu_fallback():
    m = ConvBNAct(32, 32, 1, 1)
    with pytest.raises(yolov4_amd.Y4Error):
        m(torch.zeros(1, 32, 4, 4))
    lay = YOLOLayer(recipe.MODEL_CFG, 0).eval()
    with pytest.raises(yolov4_amd.Y4Error):
        lay(torch.zeros(1, 255, 4, 4))
    if not torch.cuda.is_available():
        with pytest.raises(yolov4_amd.Y4Error):
            utils.postprocess(torch.zeros(1, 10, 85), 80)
        with pytest.raises(yolov4_amd.Y4Error):
            utils.nms(np.zeros((3, 4), np.float32), 0.5)
    # the product package must not import the oracle
    import sys
    for name, mod in list(sys.modules.items()):
        if name.startswith('yolov4_amd') and mod is not None and getattr(mod, '__file__', None):
            src = open(mod.__file__).read()
            assert 'import oracle' not in src and 'from oracle' not in src, name


def test_build_factories():
    import argparse
    m = build_model(argparse.Namespace(channels_last=True), recipe.FULL_CFG, device=torch.device('cpu'))
    assert isinstance(m, YOLOv4)
    c = build_criterion(recipe.FULL_CFG, device=torch.device('cpu'))
    assert isinstance(c, YOLOLoss) and abs(c.ignore_thresh - 0.7) < 1e-12


def test_nhwc_pitch_helper():
    t = torch.empty((2, 64, 5, 7), memory_format=torch.channels_last)
    assert ops.nhwc_pitch(t) == 64
    assert ops.nhwc_pitch(t[:, 16:48]) == 64
    assert ops.nhwc_pitch(torch.empty(2, 64, 5, 7)) is None
    assert ops.nhwc_pitch(torch.empty((2, 64, 1, 1), memory_format=torch.channels_last)) == 64
    v, ld = ops.as_nhwc(torch.arange(2 * 6 * 3 * 3, dtype=torch.float32).reshape(2, 6, 3, 3))
    assert ld == 8 and v.shape == (2, 6, 3, 3)
    assert torch.equal(v, torch.arange(2 * 6 * 3 * 3, dtype=torch.float32).reshape(2, 6, 3, 3))
