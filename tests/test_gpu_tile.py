# -*- coding: utf-8 -*-
"""csrc/conv_tile.hip: 3x3 stride-1 layers with 32 / 64 gathered channels and <= 64 produced channels on maps of >= 100 x 100
pixels run on the 2-D tile kernel (every input element staged once for all nine taps), forward and dgrad.  Checked against
torch fp64 at the fp32-grade bound, on maps whose sides are not multiples of the 8 x 16 tile, with several images per block,
and through the BatchNorm-statistics entry point (column sums accumulated over all tiles of a persistent block)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import recipe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    return torch.device('cuda:0')


def cl(t, dev):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


CASES = [
    # B, Cin, Cout, H, W
    (1, 32, 64, 104, 120),      # forward <32, 64>; dgrad <64, 32>
    (3, 64, 64, 100, 101),      # forward / dgrad <64, 32>, two N tiles forward; odd width: partial tiles on both sides
    (2, 32, 32, 129, 100),      # N = 32 < BN
    (70, 64, 32, 100, 104),     # more spatial tiles than one round of blocks, many images
    (1, 32, 36, 112, 112),      # N not a multiple of 16
]


@pytest.mark.parametrize('case', CASES)
def test_tile_kernel_forward_dgrad_and_statistics(dev, case):
    from yolov4_amd import ops
    B, ci, co, H, W = case
    x = recipe.randn((B, ci, H, W), 31)
    w = recipe.randn((co, ci, 3, 3), 32, 1.0 / np.sqrt(ci * 9))
    dy = recipe.randn((B, co, H, W), 33)
    xd, wd, dyd = cl(x, dev), cl(w, dev), cl(dy, dev)
    y = ops.conv_fwd_raw(xd, wd, 3, 1)
    assert ops.last_conv_kernel().startswith('conv3x3_tile_f16x2'), 'the tile kernel did not run'
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    err = float((y.double().cpu() - ref).abs().max())
    assert err <= 1e-5 * float(ref.abs().max()), err
    if co % 32 == 0:
        dx = ops.conv_dgrad_raw(dyd, wd, (B, ci, H, W), 3, 1)
        assert ops.last_conv_kernel().startswith('conv3x3_tile_f16x2')
        dref = torch.nn.grad.conv2d_input((B, ci, H, W), w.double(), dy.double(), 1, 1)
        err = float((dx.double().cpu() - dref).abs().max())
        assert err <= 1e-5 * float(dref.abs().max()), err
        # with the skip gradient added in the epilogue
        res = cl(recipe.randn((B, ci, H, W), 34), dev)
        dx2 = ops.conv_dgrad_raw(dyd, wd, (B, ci, H, W), 3, 1, residual=res)
        assert float((dx2 - (dx + res)).abs().max()) <= 1e-6 * float(dref.abs().max())
        # filter planes prepared by the forward call (mirrored for this kernel): bit-identical
        buf = ops.dgrad_filter_buffer(ci, co, 3, dev)
        rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
        ops.conv_fwd_bnstats_raw(xd, wd, 3, 1, rm, rv, None, 0.1, 1e-5, dgrad_filter=buf)
        dx3 = ops.conv_dgrad_raw(dyd, wd, (B, ci, H, W), 3, 1, prepared=buf)
        assert torch.equal(dx3, dx)
    if co % 4 == 0:
        rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
        y2, mean, invstd = ops.conv_fwd_bnstats_raw(xd, wd, 3, 1, rm, rv, None, 0.1, 1e-5)
        assert torch.equal(y2, y)
        yy = ref.permute(0, 2, 3, 1).reshape(-1, co)
        assert torch.allclose(mean.double().cpu(), yy.mean(0), rtol=1e-5, atol=1e-6)
        assert torch.allclose(invstd.double().cpu(), (yy.var(0, unbiased=False) + 1e-5).rsqrt(), rtol=1e-5)


def test_tile_kernel_is_deterministic_and_image_independent(dev):
    from yolov4_amd import ops
    x = recipe.randn((4, 32, 112, 112), 41)
    w = recipe.randn((64, 32, 3, 3), 42, 0.06)
    xd, wd = cl(x, dev), cl(w, dev)
    a = ops.conv_fwd_raw(xd, wd, 3, 1)
    b = ops.conv_fwd_raw(xd, wd, 3, 1)
    assert torch.equal(a, b)
    # one image alone: the same bits, provided both runs see the same operand scale (amax of the whole batch)
    amax = ops.amax_raw(xd)
    one = ops.conv_fwd_raw(xd[2:3].contiguous(memory_format=torch.channels_last), wd, 3, 1, x_amax=amax)
    full = ops.conv_fwd_raw(xd, wd, 3, 1, x_amax=amax)
    assert torch.equal(one[0], full[2])


@pytest.mark.parametrize('case', [(2, 32, 64, 208, 216), (1, 32, 64, 207, 215), (65, 32, 64, 200, 200)])
def test_tile_kernel_stride2_dgrad(dev, case):
    """dgrad of a 3x3 / stride-2 conv with 64 output channels (32 -> 64 @608 in the model): one staged dy patch per tile feeds
    all four parity classes of the output.  Even and odd input sizes, more tiles than one round of blocks."""
    from yolov4_amd import ops
    B, ci, co, H, W = case
    w = recipe.randn((co, ci, 3, 3), 52, 1.0 / np.sqrt(ci * 9))
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    dy = recipe.randn((B, co, Ho, Wo), 53)
    wd, dyd = cl(w, dev), cl(dy, dev)
    dx = ops.conv_dgrad_raw(dyd, wd, (B, ci, H, W), 3, 2)
    assert ops.last_conv_kernel().startswith('conv3x3_tile_f16x2<64, 32, 16, true>'), ops.last_conv_kernel()
    ref = torch.nn.grad.conv2d_input((B, ci, H, W), w.double(), dy.double(), 2, 1)
    err = float((dx.double().cpu() - ref).abs().max())
    assert err <= 1e-5 * float(ref.abs().max()), err
    res = cl(recipe.randn((B, ci, H, W), 54), dev)
    dx2 = ops.conv_dgrad_raw(dyd, wd, (B, ci, H, W), 3, 2, residual=res)
    assert float((dx2 - (dx + res)).abs().max()) <= 1e-6 * float(ref.abs().max())
    # prepared by the forward call
    x = cl(recipe.randn((B, ci, H, W), 55), dev)
    buf = ops.dgrad_filter_buffer(ci, co, 3, dev)
    rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
    ops.conv_fwd_bnstats_raw(x, wd, 3, 2, rm, rv, None, 0.1, 1e-5, dgrad_filter=buf)
    assert torch.equal(ops.conv_dgrad_raw(dyd, wd, (B, ci, H, W), 3, 2, prepared=buf), dx)


def test_tile_kernel_on_channel_slices(dev):
    """pixel pitch > C on all three tensors: the input is a channel slice of a wider buffer, the result goes into a slice of a
    concat buffer, the skip operand of the dgrad is a slice too (zero-copy concat, as everywhere else in the library)."""
    from yolov4_amd import ops
    B, H, W = 2, 104, 112
    xw = recipe.randn((B, 96, H, W), 71)
    w = recipe.randn((64, 32, 3, 3), 72, 0.06)
    xd, wd = cl(xw, dev), cl(w, dev)
    buf = torch.zeros((B, 192, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    ops.conv_fwd_raw(xd[:, 32:64], wd, 3, 1, out=buf[:, 64:128])
    assert ops.last_conv_kernel().startswith('conv3x3_tile_f16x2')
    ref = F.conv2d(xw[:, 32:64].double(), w.double(), None, 1, 1)
    assert float((buf[:, 64:128].double().cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float(buf[:, :64].abs().max()) == 0.0 and float(buf[:, 128:].abs().max()) == 0.0
    dyw = cl(recipe.randn((B, 128, H, W), 73), dev)
    resw = cl(recipe.randn((B, 64, H, W), 74), dev)
    dx = ops.conv_dgrad_raw(dyw[:, 64:128], wd, (B, 32, H, W), 3, 1, residual=resw[:, 32:64])
    assert ops.last_conv_kernel().startswith('conv3x3_tile_f16x2')
    dref = torch.nn.grad.conv2d_input((B, 32, H, W), w.double(), dyw[:, 64:128].double().cpu(), 1, 1) + resw[:, 32:64].double().cpu()
    assert float((dx.double().cpu() - dref).abs().max()) <= 1e-5 * float(dref.abs().max())


# ---------------------------------------------------------------- csrc/wgrad_tile.hip: filter gradient of the same family of layers
WGRAD_CASES = [
    # B, Cin, Cout, stride, H, W   (input map)
    (2, 32, 64, 1, 112, 112),
    (1, 64, 64, 1, 104, 120),      # two 32-channel input groups
    (3, 32, 64, 1, 101, 107),      # odd sides: partial tiles right and below
    (2, 32, 64, 2, 208, 208),      # stride 2: patch de-interleaved by column parity
    (1, 32, 64, 2, 202, 230),      # stride 2, output 101 x 115: partial tiles
    (70, 64, 64, 1, 100, 104),     # more tiles than one block's share many times over, many images
]


@pytest.mark.parametrize('case', WGRAD_CASES)
def test_tile_wgrad_matches_torch_and_the_split_k_kernel(dev, case, monkeypatch):
    from yolov4_amd import ops
    B, ci, co, s, H, W = case
    x = recipe.randn((B, ci, H, W), 41)
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    dy = recipe.randn((B, co, Ho, Wo), 42)
    xd, dyd = cl(x, dev), cl(dy, dev)
    dw = ops.conv_wgrad_raw(xd, dyd, (co, ci, 3, 3), 3, s)
    torch.cuda.synchronize()
    assert ops.last_conv_kernel().startswith('wgrad_tile_f16x2'), ops.last_conv_kernel()
    ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 3, 3), dy.double(), s, 1)
    err = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 3e-6, err
    # same arithmetic as the split-K kernel it replaces (same scales, same pieces): equal to accumulation order
    monkeypatch.setenv('Y4_NO_TILE_WGRAD', '1')
    dw2 = ops.conv_wgrad_raw(xd, dyd, (co, ci, 3, 3), 3, s)
    # (the switch is read once per process: if this process already used the tile kernel, the second call is the same kernel)
    assert float((dw - dw2).abs().max()) <= 3e-6 * float(ref.abs().max())
    # deterministic: fixed tile order per block, fixed-order slab reduce
    dw3 = ops.conv_wgrad_raw(xd, dyd, (co, ci, 3, 3), 3, s)
    if ops.last_conv_kernel().startswith('wgrad_tile_f16x2'):
        assert torch.equal(dw3, ops.conv_wgrad_raw(xd, dyd, (co, ci, 3, 3), 3, s))


def test_tile_wgrad_on_channel_slices_and_wide_dynamic_range(dev):
    """x a channel slice of a wider NHWC buffer (pitch 96), dy with pitch 128; x of order 1e4 with a 1e8 outlier, dy of order
    1e-6: the per-tensor power-of-two scales keep 22 significant bits (as the other f16x2 kernels)."""
    from yolov4_amd import ops
    B, ci, co, H, W = 2, 32, 64, 112, 104
    xw = torch.randn(B, 96, H, W, device=dev).contiguous(memory_format=torch.channels_last) * 1e4
    xw[0, 40, 5, 7] = 1e8
    x = xw[:, 32:64]
    dyw = torch.randn(B, 128, H, W, device=dev).contiguous(memory_format=torch.channels_last) * 1e-6
    dy = dyw[:, 64:128]
    dw = ops.conv_wgrad_raw(x, dy, (co, ci, 3, 3), 3, 1)
    torch.cuda.synchronize()
    assert ops.last_conv_kernel().startswith('wgrad_tile_f16x2'), ops.last_conv_kernel()
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (co, ci, 3, 3), dy.double().cpu(), 1, 1)
    # the outlier costs the other elements of x their low bits (scale follows the maximum): bound relative to |x|max |dy| sums
    bound = 1e8 * 2.0 ** -22 * float(dy.abs().sum(dim=(0, 2, 3)).max())
    assert float((dw.double().cpu() - ref).abs().max()) <= max(3e-6 * float(ref.abs().max()), bound)
