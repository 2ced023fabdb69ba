# -*- coding: utf-8 -*-
"""GPU parity: the HIP path (through the C ABI) against the oracle on the same
seeded inputs and against the golden fixtures the reference produced.

Tolerances (north_star): bit-exact for anchor indices, masks and NMS survivor
sets; abs 1e-4 on per-cell fp32 tensors (scaled by the tensor's magnitude for
unbounded ones: boxes in pixels, conv outputs, gradients); rel 1e-4 on the loss
scalar (an un-normalised fp32 sum).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import recipe
from oracle import head as H
from oracle import network as NW

pytestmark = pytest.mark.gpu

CFG = recipe.MODEL_CFG


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    import yolov4_amd
    assert yolov4_amd.lib().y4_device_count() >= 1, 'no gfx950 device visible to libyolov4_amd.so'
    return torch.device('cuda:0')


def close(a, b, atol=1e-4, rtol=1e-4, scale=True):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    a = a.astype(np.float64); b = b.astype(np.float64)
    s = max(1.0, float(np.abs(b).max())) if scale else 1.0
    np.testing.assert_allclose(a, b, atol=atol * s, rtol=rtol)


def tight(a, ref64, rel=1e-5):
    """fp32-grade bound for raw conv results: max |a - ref| <= rel * max |ref| against an fp64 reference (`close` scales
    its atol by the tensor's range and so accepts ~4e-4 absolute on O(1) outputs -- VERDICT r2 weak #3; this is 40x tighter:
    a k-ordered fp32 fma chain sits at ~1e-6 of the range, plain bf16 / fp16 operands at ~3e-3)."""
    a = a.detach().cpu().double()
    err = float((a - ref64).abs().max())
    bound = rel * float(ref64.abs().max())
    assert err <= bound, (err, bound)


def cl(t, dev):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------ raw conv kernels
CONV_CASES = [
    # B, Cin, Cout, k, s, H, W
    (2, 32, 64, 3, 1, 9, 9),
    (3, 64, 128, 3, 2, 13, 13),       # odd size, stride 2
    (2, 128, 128, 1, 1, 12, 12),
    (1, 256, 255, 3, 1, 10, 10),      # head conv: Cout = 255 (masked column)
    (2, 64, 32, 1, 1, 20, 20),        # Cout = 32 tile
    (2, 32, 64, 3, 2, 16, 16),
    (5, 512, 256, 1, 1, 7, 7),        # M = 245 (not a multiple of the 128-row tile)
    (2, 2048, 512, 1, 1, 5, 5),       # SPP conv2 depth
    (1, 32, 32, 3, 1, 40, 40),
]


@pytest.fixture(params=['bf16x3', 'f32', 'f16x2'])
def conv_mode(request):
    import yolov4_amd
    old = yolov4_amd.get_conv_mode()
    yolov4_amd.set_conv_mode(request.param)
    yield request.param
    yolov4_amd.set_conv_mode(old)


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, case, conv_mode):
    from yolov4_amd import ops
    B, Cin, Cout, k, s, Hh, Ww = case
    x = recipe.randn((B, Cin, Hh, Ww), 1)
    w = recipe.randn((Cout, Cin, k, k), 2, 1.0 / np.sqrt(Cin * k * k))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, (k - 1) // 2)
    gy = recipe.randn(tuple(yr.shape), 3)
    yr.backward(gy)
    y = ops.conv_fwd_raw(cl(x, dev), cl(w, dev), k, s)
    close(y, yr)
    dx = ops.conv_dgrad_raw(cl(gy, dev), cl(w, dev), (B, Cin, Hh, Ww), k, s)
    close(dx, xr.grad)
    dw = ops.conv_wgrad_raw(cl(x, dev), cl(gy, dev), (Cout, Cin, k, k), k, s)
    close(dw, wr.grad)
    # and tightly, against fp64 (all three modes are fp32-grade)
    x64, w64, g64 = x.double().requires_grad_(True), w.double().requires_grad_(True), gy.double()
    y64 = F.conv2d(x64, w64, None, s, (k - 1) // 2)
    y64.backward(g64)
    tight(y, y64.detach())
    tight(dx, x64.grad)
    tight(dw, w64.grad)


def test_conv_fused_epilogue(dev, conv_mode):
    from yolov4_amd import ops
    B, Cin, Cout, k, s, Hh = 2, 64, 64, 3, 1, 11
    x = recipe.randn((B, Cin, Hh, Hh), 4)
    w = recipe.randn((Cout, Cin, k, k), 5, 0.05)
    sc = recipe.rand((Cout,), 6) + 0.5
    sh = recipe.randn((Cout,), 7)
    res = recipe.randn((B, Cout, Hh, Hh), 8)
    for act, fn in (('mish', NW.mish), ('leaky_relu', lambda t: F.leaky_relu(t, 0.1)), ('relu', F.relu), ('linear', lambda t: t)):
        ref = fn(F.conv2d(x, w, None, s, 1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res
        got = ops.conv_fwd_raw(cl(x, dev), cl(w, dev), k, s, sc.to(dev), sh.to(dev), act, cl(res, dev))
        close(got, ref)


def test_conv_output_into_channel_slice(dev):
    """pixel pitch > C: conv reads a channel slice and writes a channel slice (zero-copy concat)"""
    from yolov4_amd import ops
    x = recipe.randn((2, 96, 8, 8), 9)
    w = recipe.randn((64, 32, 1, 1), 10, 0.2)
    xd = cl(x, dev)
    buf = torch.zeros((2, 128, 8, 8), device=dev).contiguous(memory_format=torch.channels_last)
    ops.conv_fwd_raw(xd[:, 32:64], cl(w, dev), 1, 1, out=buf[:, 64:128])
    close(buf[:, 64:128], F.conv2d(x[:, 32:64], w))
    assert float(buf[:, :64].abs().max()) == 0.0


def test_stem_kernels(dev):
    from yolov4_amd import ops
    x = recipe.randn((3, 3, 37, 41), 11)
    w = recipe.randn((32, 3, 3, 3), 12, 0.3)
    xr = x.clone(); wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    gy = recipe.randn(tuple(yr.shape), 13)
    yr.backward(gy)
    for xin in (x.to(dev), cl(x, dev)):                      # NCHW as the reference feeds it, and NHWC
        close(ops.conv_fwd_raw(xin, cl(w, dev), 3, 1), yr)
        close(ops.conv_wgrad_raw(xin, cl(gy, dev), (32, 3, 3, 3), 3, 1), wr.grad)


def test_conv_linearity_at_full_size(dev, conv_mode):
    """Size-independent property at a BASELINE-size layer (128->128 3x3 @76x76, B=8):
    conv(a*x1 + x2) == a*conv(x1) + conv(x2) within fp32 rounding."""
    from yolov4_amd import ops
    g = torch.Generator(device='cpu'); g.manual_seed(5)
    x1 = torch.randn((8, 128, 76, 76), generator=g); x2 = torch.randn((8, 128, 76, 76), generator=g)
    w = cl(torch.randn((128, 128, 3, 3), generator=g) * 0.03, dev)
    y1 = ops.conv_fwd_raw(cl(x1, dev), w, 3, 1); y2 = ops.conv_fwd_raw(cl(x2, dev), w, 3, 1)
    y12 = ops.conv_fwd_raw(cl(2.5 * x1 + x2, dev), w, 3, 1)
    close(y12, 2.5 * y1 + y2, 2e-5, 1e-4)
    # spot-check one output row against the CPU reference
    ref = F.conv2d(x1[:1], w.cpu(), None, 1, 1)
    close(y1[:1], ref)


def test_conv_tensors_beyond_4gib(dev, conv_mode):
    """Maximum sizes: a 6 GiB activation (B = 64, 64 ch @608x608).  Kernels address through 32-bit buffer windows
    re-based per block, so the full-batch launch must equal the same kernels run on < 4 GiB sub-batches."""
    from yolov4_amd import ops
    B, Cin, Cout, Hh = 64, 64, 32, 608
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn((B, Cin, Hh, Hh), device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    assert x.numel() * 4 > (1 << 32)
    w = (torch.randn((Cout, Cin, 1, 1), device=dev, generator=g) * 0.1).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd_raw(x, w, 1, 1)
    same = torch.equal if conv_mode != 'f16x2' else (lambda a, b: bool(((a - b).abs() <= 2e-6 * b.abs().max()).all()))
    # (f16x2 scales every operand by a power of two taken from the tensor's maximum: a sub-batch with a smaller
    # maximum is split at a finer grid, so results agree to the 2^-22 split error, not to the bit)
    for b0 in (0, 40):
        assert same(y[b0:b0 + 24], ops.conv_fwd_raw(x[b0:b0 + 24], w, 1, 1))
    # dgrad produces the > 4 GiB tensor, wgrad reduces over it
    dy = torch.randn((B, Cout, Hh, Hh), device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    dx = ops.conv_dgrad_raw(dy, w, (B, Cin, Hh, Hh), 1, 1)
    assert same(dx[40:64], ops.conv_dgrad_raw(dy[40:64], w, (24, Cin, Hh, Hh), 1, 1))
    dw = ops.conv_wgrad_raw(x, dy, (Cout, Cin, 1, 1), 1, 1)
    parts = sum(ops.conv_wgrad_raw(x[b0:b0 + 16], dy[b0:b0 + 16], (Cout, Cin, 1, 1), 1, 1).double() for b0 in range(0, 64, 16))
    close(dw, parts, 1e-5, 1e-4)


@pytest.mark.parametrize('B,ci,co,H', [(3, 64, 64, 211), (2, 32, 64, 300), (2, 64, 128, 270), (2, 64, 32, 301), (2, 64, 96, 270),
                                       (2, 128, 128, 270), (2, 128, 64, 263), (2, 128, 32, 257)])
def test_conv1x1_streaming_kernel(dev, B, ci, co, H):
    """1x1 layers with K <= 128, N <= 128 and M >= 131072 rows take the persistent streaming kernel (forward, BN-statistics
    forward and dgrad); M is not a multiple of the 128-row tile here and N = 96 leaves a partly masked column tile."""
    from yolov4_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + ci + co + H)
    x = cl(torch.randn((B, ci, H, H), generator=g), dev)
    w = cl(torch.randn((co, ci, 1, 1), generator=g) * 0.1, dev)
    dy = cl(torch.randn((B, co, H, H), generator=g), dev)
    ref = F.conv2d(x.double(), w.double())
    close(ops.conv_fwd_raw(x, w, 1, 1), ref, 1e-5, 1e-5)
    close(ops.conv_dgrad_raw(dy, w, (B, ci, H, H), 1, 1), F.conv_transpose2d(dy.double(), w.double()), 1e-5, 1e-5)
    rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
    nbt = torch.zeros((), dtype=torch.long, device=dev)
    y, mean, invstd = ops.conv_fwd_bnstats_raw(x, w, 1, 1, rm, rv, nbt, 0.1, 1e-5)
    close(y, ref, 1e-5, 1e-5)
    close(mean, ref.mean(dim=(0, 2, 3)), 1e-6, 1e-5, scale=False)
    close(invstd, (ref.var(dim=(0, 2, 3), unbiased=False) + 1e-5).rsqrt(), 1e-6, 1e-5, scale=False)
    assert int(nbt) == 1


def test_conv_non_finite_inputs_stay_non_finite(dev, conv_mode):
    """A NaN or Inf activation must not be laundered into a finite number by the operand split (bf16x3: x - trunc(x)
    of an Inf is NaN, so an Inf input surfaces as NaN where the fp32 chain gives +-Inf -- both are non-finite)."""
    from yolov4_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 64, 12, 12), generator=g)
    x[0, 3, 5, 5] = float('nan')
    x[0, 7, 2, 9] = float('inf')
    w = torch.randn((32, 64, 3, 3), generator=g) * 0.1
    y = ops.conv_fwd_raw(cl(x, dev), cl(w, dev), 3, 1).cpu()
    ref = F.conv2d(x, w, padding=1)
    assert torch.equal(torch.isfinite(y), torch.isfinite(ref))
    assert torch.isnan(y[0, :, 4:7, 4:7]).all()
    close(torch.nan_to_num(y, 0.0, 0.0, 0.0), torch.nan_to_num(ref, 0.0, 0.0, 0.0), 1e-5, 1e-5)


def test_conv_plain_bf16_mode(dev):
    """BASELINE config 5 arithmetic (bf16 MFMA operands, fp32 accumulate): mixed precision, so only a loose
    bound holds against the fp32 reference -- 2^-8 relative per operand -> ~1e-2 of the output range."""
    import yolov4_amd
    from yolov4_amd import ops
    old = yolov4_amd.get_conv_mode()
    yolov4_amd.set_conv_mode('bf16')
    try:
        for (B, Cin, Cout, k, s, Hh) in [(2, 128, 128, 3, 1, 12), (2, 64, 255, 1, 1, 9), (2, 64, 128, 3, 2, 13)]:
            x = recipe.randn((B, Cin, Hh, Hh), 1); w = recipe.randn((Cout, Cin, k, k), 2, 1.0 / np.sqrt(Cin * k * k))
            xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
            yr = F.conv2d(xr, wr, None, s, (k - 1) // 2)
            gy = recipe.randn(tuple(yr.shape), 3)
            yr.backward(gy)
            close(ops.conv_fwd_raw(cl(x, dev), cl(w, dev), k, s), yr, 2e-2, 2e-2)
            close(ops.conv_dgrad_raw(cl(gy, dev), cl(w, dev), (B, Cin, Hh, Hh), k, s), xr.grad, 2e-2, 2e-2)
            close(ops.conv_wgrad_raw(cl(x, dev), cl(gy, dev), (Cout, Cin, k, k), k, s), wr.grad, 2e-2, 2e-2)
    finally:
        yolov4_amd.set_conv_mode(old)


def test_conv_modes_accuracy_vs_fp64(dev):
    """Both conv arithmetics against an fp64 convolution (K = 4608): the split-bf16 mode must be
    fp32-grade, i.e. no worse than 1.5x the error of the exact fp32 fma chain."""
    import yolov4_amd
    from yolov4_amd import ops
    x = recipe.randn((2, 512, 19, 19), 41)
    w = recipe.randn((256, 512, 3, 3), 42, 1.0 / np.sqrt(4608))
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    err = {}
    old = yolov4_amd.get_conv_mode()
    for mode in ('f32', 'bf16x3', 'f16x2'):
        yolov4_amd.set_conv_mode(mode)
        y = ops.conv_fwd_raw(cl(x, dev), cl(w, dev), 3, 1).double().cpu()
        err[mode] = float(((y - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    yolov4_amd.set_conv_mode(old)
    print('rms error / rms output vs fp64 at K = 4608:', err)
    assert err['f32'] < 5e-6 and err['bf16x3'] < 5e-6 and err['f16x2'] < 5e-6, err
    assert err['bf16x3'] <= 1.5 * err['f32'] and err['f16x2'] <= 1.5 * err['f32'], err


# ------------------------------------------------------------------ f16x2: behaviour over the tensor's dynamic range
# conv_f16x2.hip states: s x = hi + 2^-11 lo + e with |e| <= 2^-22 |s x| for |x| >= 2^-29 max|T| (floor 2^-50 max|T|),
# products exact, lo*lo (2^-22) dropped, fp32 accumulation.  Per output cell that is an error of a few 2^-22 of
# (|x| (*) |w|)(cell), the convolution of the absolute values -- a CELL-WISE bound, unlike `close()`, whose tolerance
# scales with the tensor's largest value and would hide a lost operand next to an outlier.
def _cellwise_bound_ok(y, ref64, absconv64, c=16.0, floor=0.0):
    err = (y.double().cpu() - ref64).abs()
    bound = c * 2.0 ** -22 * absconv64 + floor
    bad = err > bound
    assert not bool(bad.any()), (int(bad.sum()), float((err / (absconv64 + 1e-300)).max()))
    return float((err / (absconv64 + 1e-300)).max())


def _f16x2(fn):
    import yolov4_amd
    old = yolov4_amd.get_conv_mode()
    yolov4_amd.set_conv_mode('f16x2')
    try:
        return fn()
    finally:
        yolov4_amd.set_conv_mode(old)


def test_f16x2_one_huge_outlier_among_unit_activations(dev):
    """(i) one 1e8 activation among O(1) values: the scale follows the outlier (2^-27 of it is still inside the
    full-precision range 2^-29), so cells that never see the outlier keep fp32-grade accuracy."""
    from yolov4_amd import ops
    x = recipe.randn((2, 64, 20, 20), 51)
    x[1, 17, 9, 11] = 1.0e8
    w = recipe.randn((96, 64, 3, 3), 52, 1.0 / np.sqrt(576))
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    ab = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
    worst = _f16x2(lambda: _cellwise_bound_ok(ops.conv_fwd_raw(cl(x, dev), cl(w, dev), 3, 1), ref, ab))
    print('outlier: worst cell error / (|x| * |w|) =', worst)
    # the same through the producer-side maximum (an upper bound is as good as the exact one)
    xd = cl(x, dev)
    cell = torch.tensor([np.float32(3.0e8).view(np.int32)], dtype=torch.int32, device=dev)
    _f16x2(lambda: _cellwise_bound_ok(ops.conv_fwd_raw(xd, cl(w, dev), 3, 1, x_amax=cell), ref, ab))


def test_f16x2_sparse_unit_gradient_over_a_tiny_background(dev):
    """(ii) dy with 1 % O(1) cells and the rest 1e-9 * randn (what a detection loss produces: a few positives, a sea of
    almost-zero objectness gradients), through dgrad and wgrad."""
    from yolov4_amd import ops
    B, Ci, Co, H = 2, 64, 96, 20
    g = torch.Generator().manual_seed(53)
    dy = 1e-9 * torch.randn((B, Co, H, H), generator=g)
    hot = torch.rand((B, Co, H, H), generator=g) < 0.01
    dy[hot] = torch.randn(int(hot.sum()), generator=g)
    x = recipe.randn((B, Ci, H, H), 54)
    w = recipe.randn((Co, Ci, 3, 3), 55, 1.0 / np.sqrt(576))
    dx_ref = F.conv_transpose2d(dy.double(), w.double(), None, 1, 1)
    dx_ab = F.conv_transpose2d(dy.double().abs(), w.double().abs(), None, 1, 1)
    xp = F.pad(x.double(), (1, 1, 1, 1))
    dw_ref = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64)
    dw_ab = torch.zeros_like(dw_ref)
    for r in range(3):
        for q in range(3):
            patch = xp[:, :, r:r + H, q:q + H]
            dw_ref[:, :, r, q] = torch.einsum('bnhw,bchw->nc', dy.double(), patch)
            dw_ab[:, :, r, q] = torch.einsum('bnhw,bchw->nc', dy.double().abs(), patch.abs())

    def run():
        dx = ops.conv_dgrad_raw(cl(dy, dev), cl(w, dev), (B, Ci, H, H), 3, 1)
        a = _cellwise_bound_ok(dx, dx_ref, dx_ab)
        dw = ops.conv_wgrad_raw(cl(x, dev), cl(dy, dev), (Co, Ci, 3, 3), 3, 1)
        b = _cellwise_bound_ok(dw, dw_ref, dw_ab)
        return a, b
    print('sparse gradient: worst dgrad / wgrad cell error / (|a| * |b|) =', _f16x2(run))


def test_f16x2_all_zero_and_extreme_magnitude_operands(dev):
    """(iii) an all-zero operand (maximum 0 -> scale 1) gives exact zeros, no NaN; (iv) tensors whose maximum sits at the
    ends of the fp32 range (the scale exponent is clamped, conv_f16x2.hip f16x2_scale_exp) stay finite and accurate."""
    from yolov4_amd import ops
    w = recipe.randn((64, 32, 3, 3), 56, 1.0 / np.sqrt(288))
    x0 = torch.zeros(1, 32, 12, 12)

    def zero():
        y = ops.conv_fwd_raw(cl(x0, dev), cl(w, dev), 3, 1)
        assert bool((y == 0).all())
        y = ops.conv_fwd_raw(cl(recipe.randn((1, 32, 12, 12), 57), dev), cl(torch.zeros_like(w), dev), 3, 1)
        assert bool((y == 0).all())
    _f16x2(zero)
    for expo in (-100, -60, 60, 100):
        x = recipe.randn((1, 32, 12, 12), 58) * 2.0 ** expo
        ref = F.conv2d(x.double(), w.double(), None, 1, 1)
        ab = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
        y = _f16x2(lambda: ops.conv_fwd_raw(cl(x, dev), cl(w, dev), 3, 1))
        assert bool(torch.isfinite(y).all()), expo
        _cellwise_bound_ok(y, ref, ab)
    # subnormal maximum: no scale can lift it into fp16's range -- the documented behaviour is "taken unscaled", i.e. the
    # operand rounds to zero; the result must be finite (zeros), never NaN
    xs = recipe.randn((1, 32, 12, 12), 59) * 2.0 ** -140
    y = _f16x2(lambda: ops.conv_fwd_raw(cl(xs, dev), cl(w, dev), 3, 1))
    assert bool(torch.isfinite(y).all())


# ------------------------------------------------------------------ ConvBNAct / blocks against the reference's goldens
def _load_cba(g, name, dev):
    from yolov4_amd.darknet.darknet import ConvBNAct
    cin, cout, k, s, bn, bias, B, Hh = [int(v) for v in g[f'{name}.cfg']]
    m = ConvBNAct(cin, cout, k, s, bias=bool(bias), bn=bool(bn), act=str(g[f'{name}.act']))
    sd = {kk[len(name) + 4:]: torch.from_numpy(g[kk].copy()) for kk in g.files if kk.startswith(name + '.sd.')}
    m.load_state_dict(sd)
    return m.to(dev)


def test_convbnact_golden(dev, golden):
    g = golden('convbnact')
    for name in [str(n) for n in g['names']]:
        m = _load_cba(g, name, dev)
        x = torch.from_numpy(g[f'{name}.x'].copy()).to(dev)
        m.eval()
        with torch.no_grad():
            close(m(x), g[f'{name}.eval_y'])
        m.train()
        is_stem = x.shape[1] == 3
        xin = x.clone().requires_grad_(not is_stem)
        y = m(xin)
        close(y, g[f'{name}.train_y'])
        y.backward(torch.from_numpy(g[f'{name}.gy'].copy()).to(dev))
        if not is_stem:
            close(xin.grad, g[f'{name}.gx'])
        for kk in g.files:
            if kk.startswith(name + '.grad.'):
                p = dict(m.named_parameters())[kk[len(name) + 6:]]
                close(p.grad, g[kk], 2e-4, 1e-3)
            if kk.startswith(name + '.after.'):
                close(m.state_dict()[kk[len(name) + 7:]], g[kk], 1e-5, 1e-4)


def _fill(mod, seed, dev):
    sd = mod.state_dict()
    recipe.fill_state_dict_(sd, seed)
    mod.load_state_dict(sd)
    return mod.to(dev)


def test_blocks_golden(dev, golden):
    from yolov4_amd.darknet.darknet import CSPDownSample, CSPDownSample0, ResBlock
    from yolov4_amd.yolo.model.yolov4 import SPPBlock, Upsample
    g = golden('blocks')
    cases = [('resblock', ResBlock(32, num_blocks=2), 601), ('csp0', CSPDownSample0(32, 64), 602),
             ('csp', CSPDownSample(32, 64, num_blocks=2), 603), ('spp', SPPBlock(), 604)]
    for name, mod, seed in cases:
        mod = _fill(mod, seed, dev)
        x = torch.from_numpy(g[f'{name}.x'].copy()).to(dev)
        mod.train()
        xin = x.clone().requires_grad_(True)
        y = mod(xin)
        close(y, g[f'{name}.train_y'])
        y.backward(recipe.randn(tuple(y.shape), seed + 1).to(dev))
        close(xin.grad, g[f'{name}.gx'], 2e-4, 1e-3)
        named = dict(mod.named_parameters())
        for kk, ref in zip([str(q) for q in g[f'{name}.gradnorm_keys']], g[f'{name}.gradnorm']):
            got = float(named[kk].grad.double().norm())
            assert abs(got - ref) <= 1e-3 * max(ref, 1e-3), (name, kk, got, ref)
        mod.eval()
        with torch.no_grad():
            close(mod(x), g[f'{name}.eval_y'])
    up = Upsample()
    ux = torch.from_numpy(g['up.x'].copy()).to(dev)
    close(up(ux, (2, 8, 6, 6)), g['up.train'], 0, 0)
    uxg = ux.clone().requires_grad_(True)
    gy = recipe.randn((2, 8, 6, 6), 99)
    up(uxg, (2, 8, 6, 6)).backward(gy.to(dev))
    close(uxg.grad, gy.view(2, 8, 3, 2, 3, 2).sum((3, 5)), 1e-6, 1e-6)


def test_bn_statistics_large_mean(dev):
    """E[y^2]-E[y]^2 is formed in fp64: a mean 30x the std must not destroy the variance."""
    from yolov4_amd import ops
    y = recipe.randn((8, 64, 40, 40), 21) * 0.5 + 15.0
    mean, invstd = ops.bn_stats_raw(cl(y, dev), None, None, None, 0.1, 1e-5)
    ref_m = y.double().mean((0, 2, 3)); ref_v = y.double().var((0, 2, 3), unbiased=False)
    close(mean, ref_m, 1e-5, 1e-6)
    close(invstd, 1.0 / torch.sqrt(ref_v + 1e-5), 1e-4, 1e-4)


@pytest.mark.parametrize('case', [(2, 64, 410, 410, 'mish', False), (1, 32, 700, 610, 'leaky_relu', False), (3, 128, 200, 233, 'mish', True),
                                  (64, 512, 19, 19, 'leaky_relu', True)])
def test_bn_backward_sweeps_cover_every_row_at_large_sizes(dev, case):
    """BatchNorm + activation backward (two sweeps + folds) on tensors large enough that a block's share of rows is NOT a
    multiple of the 4 rows it keeps in flight (the small fixtures never are): dy, dgamma, dbeta against torch fp64, fp32 and
    plane output.  (A row order that left the last partial group of every block unvisited passed every small test and was
    caught only by the batch-permutation test of the whole network.)"""
    from yolov4_amd import ops
    B, C, H, W, act, planes = case
    y = recipe.randn((B, C, H, W), 61) * 1.3 + 0.2
    dz = recipe.randn((B, C, H, W), 62)
    gamma = recipe.rand((C,), 63) + 0.5
    beta = recipe.randn((C,), 64) * 0.1
    y64 = y.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    u = F.batch_norm(y64, None, None, g64, b64, True, 0.1, 1e-5)
    z = NW.mish(u) if act == 'mish' else F.leaky_relu(u, 0.1)
    z.backward(dz.double())
    yd, dzd = cl(y, dev), cl(dz, dev)
    mean, invstd = ops.bn_stats_raw(yd, None, None, None, 0.1, 1e-5)
    cell = ops.planes_cell(dev, 8) if planes else None
    dy, dgamma, dbeta = ops.bn_act_bwd_raw(dzd, yd, mean, invstd, gamma.to(dev), beta.to(dev), act, planes=cell)
    torch.cuda.synchronize()
    scale = float(y64.grad.abs().max())
    assert float((dgamma.double().cpu() - g64.grad).abs().max()) <= 2e-4 * float(g64.grad.abs().max())
    assert float((dbeta.double().cpu() - b64.grad).abs().max()) <= 2e-4 * float(b64.grad.abs().max())
    if planes:
        # [pixel][C / 32][hi 32 halfs | lo 32 halfs], scaled by the power of two of the bound in cell[5]
        mem = dy.permute(0, 2, 3, 1).contiguous().cpu().numpy()          # NHWC: the bytes as they lie in HBM, 4 per element
        raw = mem.view(np.float16).reshape(B, H, W, C // 32, 2, 32).astype(np.float64)
        e8 = (int(cell[5].item()) >> 23) & 0xff
        s = 2.0 ** (268 - e8 - 127)
        val = (raw[..., 0, :] + raw[..., 1, :] / 2048.0) / s
        got = torch.from_numpy(val.reshape(B, H, W, C)).permute(0, 3, 1, 2)
    else:
        got = dy.double().cpu()
    assert float((got - y64.grad).abs().max()) <= 2e-5 * scale


# ------------------------------------------------------------------ YOLO head
@pytest.mark.parametrize('l', [0, 1, 2])
def test_yololayer_golden(dev, golden, l):
    from yolov4_amd.yolo.model.yololayer import YOLOLayer
    g = golden('yololayer')
    lay = YOLOLayer(CFG, l, device=dev)
    x = torch.from_numpy(g[f'x{l}'].copy()).to(dev)
    lay.train()
    xin = x.clone().requires_grad_(True)
    r = lay(xin)
    assert r['layer_no'] == l
    close(r['output'], g[f'train_output{l}'], 1e-6, 1e-6, scale=False)
    close(r['pred'], g[f'train_pred{l}'], 1e-5, 1e-5)
    go = recipe.randn(tuple(r['output'].shape), 200 + l).to(dev)
    gp = recipe.randn(tuple(r['pred'].shape), 300 + l).to(dev)
    ((r['output'] * go).sum() + (r['pred'] * gp).sum()).backward()
    close(xin.grad, g[f'grad_x{l}'], 1e-5, 1e-4)
    lay.eval()
    with torch.no_grad():
        close(lay(x), g[f'eval_out{l}'], 1e-5, 1e-5)


def test_yololoss_golden(dev, golden):
    from yolov4_amd.yolo.model.yololayer import YOLOLayer
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    g = golden('yololoss')
    crit = YOLOLoss(CFG, ignore_thresh=0.7, device=dev)
    labels = torch.from_numpy(g['labels'].copy())                      # float64 [B,60,5], as Transform emits
    xs, outs = [], []
    for l in range(3):
        x = torch.from_numpy(g[f'logits{l}'].copy()).to(dev).requires_grad_(True)
        lay = YOLOLayer(CFG, l, device=dev).train()
        r = lay(x)
        xs.append(x); outs.append(r)
        tgt, obj, tm, ts = crit.build_target(r['output'], r['pred'], l, labels)
        assert torch.equal(obj.cpu(), torch.from_numpy(g[f'obj_mask{l}']))          # bit-exact masks
        assert torch.equal(tm[..., 0].cpu(), torch.from_numpy(g[f'tgt_mask{l}']))
        assert bool((tm == tm[..., :1]).all())
        assert torch.equal((tgt != 0).cpu(), torch.from_numpy(g[f'target{l}'] != 0))  # anchor / cell / class indices
        close(tgt, g[f'target{l}'], 1e-6, 1e-6, scale=False)
        close(ts, g[f'tgt_scale{l}'], 1e-6, 1e-6, scale=False)
    for l in range(3):
        lay = YOLOLayer(CFG, l, device=dev).train()
        r = lay(torch.from_numpy(g[f'logits{l}'].copy()).to(dev))
        ll = float(crit([r], {'padded_labels': labels}))
        ref = float(g[f'loss_layer{l}'])
        assert abs(ll - ref) <= 1e-4 * abs(ref), (l, ll, ref)
    loss = crit(outs, {'padded_labels': labels})
    ref = float(g['loss'])
    assert abs(float(loss) - ref) <= 1e-4 * abs(ref)
    loss.backward()
    for l in range(3):
        close(xs[l].grad, g[f'grad_logits{l}'], 1e-5, 1e-4)
        close(outs[l]['output'], g[f'mutated_output{l}'], 1e-6, 1e-6, scale=False)   # the in-place side effect


def test_head_and_loss_full_size_vs_oracle(dev):
    """BASELINE-size heads (608 px: F = 76/38/19, B = 4, synthetic labels per SURVEY 8d incl. an empty image):
    decode + target assignment + loss + gradient against the oracle on the same logits; masks bit-exact."""
    from yolov4_amd.yolo.model.yololayer import YOLOLayer
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    B = 4
    labels = recipe.synth_labels(B, 608, 91, counts=[60, 0, 17, 1])
    crit = YOLOLoss(CFG, ignore_thresh=0.7, device=dev, mutate_outputs=False)
    xs, outs, logits = [], [], []
    for l, Fs in enumerate((76, 38, 19)):
        lg = recipe.synth_head_logits(B, Fs, 700 + l)
        logits.append(lg.numpy())
        x = lg.to(dev).requires_grad_(True)
        r = YOLOLayer(CFG, l, device=dev).train()(x)
        xs.append(x); outs.append(r)
        o_ref, p_ref = H.yolo_decode(lg.numpy(), l, CFG, True)
        close(r['output'], o_ref, 1e-6, 1e-6, scale=False)
        close(r['pred'], p_ref, 1e-5, 1e-5)
        # masks from the oracle on the HIP path's own decode (same inputs -> must be identical)
        _, obj_ref, tm_ref, _ = H.build_target(r['output'].detach().cpu().numpy(), r['pred'].detach().cpu().numpy(), l,
                                               labels.numpy(), CFG, 0.7)
        _, obj, tm, _ = crit.build_target(r['output'], r['pred'], l, labels)
        assert np.array_equal(obj.cpu().numpy(), obj_ref)
        assert np.array_equal(tm.cpu().numpy(), tm_ref)
    loss = crit(outs, {'padded_labels': labels})
    ref_loss, ref_grads = H.yolo_loss(logits, labels.numpy(), CFG, 0.7)
    assert abs(float(loss.detach()) - ref_loss) <= 1e-4 * abs(ref_loss)
    loss.backward()
    for l in range(3):
        close(xs[l].grad, ref_grads[l], 1e-5, 1e-4)


def test_loss_edge_cases_vs_oracle(dev):
    """Ragged / empty label tensors: every image empty (objectness-only loss), K = 7 instead of 60, a truth on the
    last cell, and a label whose class index is the last class."""
    from yolov4_amd.yolo.model.yololayer import YOLOLayer
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    B, S = 3, 128
    cases = {'all_empty': torch.zeros((B, 60, 5), dtype=torch.float64)}
    lab = torch.zeros((B, 7, 5), dtype=torch.float64)
    lab[0, 0] = torch.tensor([127.5, 127.5, 30.0, 40.0, 79.0])      # last cell of every grid, last class
    lab[0, 1] = torch.tensor([0.5, 0.5, 100.0, 90.0, 0.0])          # first cell
    lab[2, 0] = torch.tensor([64.0, 64.0, 120.0, 125.0, 17.0])      # cell boundary exactly
    cases['k7'] = lab
    for name, labels in cases.items():
        crit = YOLOLoss(CFG, ignore_thresh=0.7, device=dev, mutate_outputs=False)
        xs, outs, logits = [], [], []
        for l, Fs in enumerate((16, 8, 4)):
            lg = recipe.synth_head_logits(B, Fs, 800 + l)
            logits.append(lg.numpy())
            x = lg.to(dev).requires_grad_(True)
            xs.append(x); outs.append(YOLOLayer(CFG, l, device=dev).train()(x))
        loss = crit(outs, {'padded_labels': labels})
        ref_loss, ref_grads = H.yolo_loss(logits, labels.numpy(), CFG, 0.7)
        assert abs(float(loss.detach()) - ref_loss) <= 1e-4 * abs(ref_loss), name
        loss.backward()
        for l in range(3):
            close(xs[l].grad, ref_grads[l], 1e-5, 1e-4)
            _, obj_ref, tm_ref, _ = H.build_target(outs[l]['output'].detach().cpu().numpy(), outs[l]['pred'].detach().cpu().numpy(),
                                                   l, labels.numpy(), CFG, 0.7)
            _, obj, tm, _ = crit.build_target(outs[l]['output'], outs[l]['pred'], l, labels)
            assert np.array_equal(obj.cpu().numpy(), obj_ref) and np.array_equal(tm.cpu().numpy(), tm_ref), (name, l)


def test_postprocess_no_candidates_and_single_box(dev):
    from yolov4_amd.yolo.util.utils import postprocess
    p = torch.zeros((2, 50, 85), device=dev)
    assert postprocess(p.clone(), 80, 0.5, 0.4) == [None, None]
    p[1, 7, :5] = torch.tensor([100.0, 120.0, 40.0, 60.0, 0.9], device=dev)
    p[1, 7, 5 + 33] = 0.8
    out = postprocess(p, 80, 0.5, 0.4)
    assert out[0] is None and out[1].shape == (1, 7)
    np.testing.assert_allclose(out[1].cpu().numpy()[0], [80.0, 90.0, 120.0, 150.0, 0.9, 0.8, 33.0], rtol=1e-6)


def test_postprocess_golden(dev, golden):
    from yolov4_amd.yolo.util.utils import postprocess
    g = golden('postprocess')
    for case in range(2):
        conf, thre = [float(v) for v in g[f'params{case}']]
        for host in (False, True):
            p = torch.from_numpy(g[f'pred{case}'].copy())
            if not host:
                p = p.to(dev)
            out = postprocess(p, 80, conf, thre)
            assert torch.equal(p[:, :, :4].cpu(), torch.from_numpy(g[f'xyxy{case}']))   # in-place xyxy, exact
            for b in range(p.shape[0]):
                if bool(g[f'isnone{case}_{b}']):
                    assert out[b] is None
                else:
                    assert torch.equal(out[b].cpu(), torch.from_numpy(g[f'det{case}_{b}']))   # survivors + order: exact


def test_postprocess_many_classes_and_candidate_overflow(dev):
    """with 243..250 classes the LDS-tiled count kernel would need more than the 64-KiB default (tile + two
    histograms): the launcher must take the row-per-thread kernel instead.  And a threshold low enough for MANY classes per
    box overflows the first candidate-buffer guess (B*N/2): the call must notice on its one host read and repeat with
    room.  Both against the oracle, survivors and order exact."""
    from yolov4_amd.yolo.util.utils import postprocess
    rng = np.random.RandomState(5)
    for C, N, conf in ((250, 300, 0.3), (80, 1500, 0.002)):
        B = 2
        p = np.zeros((B, N, 5 + C), dtype=np.float32)
        p[..., 0:2] = rng.uniform(50, 550, (B, N, 2))
        p[..., 2:4] = rng.uniform(10, 200, (B, N, 2))
        p[..., 4] = rng.uniform(0.05, 1.0, (B, N))
        p[..., 5:] = rng.uniform(0, 1, (B, N, C)) ** (2 if conf < 0.01 else 6)
        ref = H.postprocess(p.copy(), C, conf, 0.45)
        got = postprocess(torch.from_numpy(p.copy()).to(dev), C, conf, 0.45)
        ncand = int(((p[..., 5:] * p[..., 4:5]) >= conf).sum())
        if C == 80:
            assert ncand > max(B * N // 2, 1 << 16), ncand          # really exercises the overflow path
        for b in range(B):
            assert (ref[b] is None) == (got[b] is None)
            if ref[b] is not None:
                assert got[b].shape == ref[b].shape
                assert np.array_equal(got[b].cpu().numpy(), ref[b])


def test_nms_golden(dev, golden):
    from yolov4_amd.yolo.util.utils import nms
    g = golden('iou_nms')
    assert np.array_equal(nms(g['box'], 0.45, score=g['score']), g['keep45'])
    assert np.array_equal(nms(g['box'], 0.3, score=g['score'], limit=7), g['keep30_lim'])
    assert np.array_equal(nms(g['box'], 0.5), g['keep_noscore'])
    k = nms(np.zeros((0, 4), np.float32), 0.5, score=np.zeros((0,), np.float32))
    assert k.shape == (0,) and k.dtype == np.int32
    # defined tie order: equal scores -> lower index first
    box = np.array([[0, 0, 10, 10], [100, 100, 110, 110], [0, 0, 10, 10]], np.float32)
    assert nms(box, 0.5, score=np.array([0.5, 0.5, 0.5], np.float32)).tolist() == [0, 1]


def test_postprocess_properties_full_size(dev):
    """BASELINE-size input (N = 22743 boxes @608): survivor sets equal the oracle's on the
    same input, NMS is idempotent, and nothing below the confidence threshold survives."""
    from yolov4_amd.yolo.util.utils import postprocess
    pred = recipe.synth_predictions(2, 22743, 31, n_clusters=150)
    ref = H.postprocess(pred.numpy().copy(), 80, 0.4, 0.45)
    out = postprocess(pred.clone().to(dev), 80, 0.4, 0.45)
    for b in range(2):
        assert np.array_equal(out[b].cpu().numpy(), ref[b])
        d = out[b]
        assert float((d[:, 4] * d[:, 5]).min()) >= 0.4
        # idempotence: feed the survivors back (as xywh) -> the same set survives
        again = torch.zeros((1, d.shape[0], 85), device=dev)
        again[0, :, 0] = (d[:, 0] + d[:, 2]) / 2; again[0, :, 1] = (d[:, 1] + d[:, 3]) / 2
        again[0, :, 2] = d[:, 2] - d[:, 0]; again[0, :, 3] = d[:, 3] - d[:, 1]
        again[0, :, 4] = d[:, 4]
        again[0, torch.arange(d.shape[0]), 5 + d[:, 6].long()] = d[:, 5]
        o2 = postprocess(again, 80, 0.4, 0.45)[0]
        assert o2.shape[0] == d.shape[0]
        assert torch.equal(o2[:, 4:], d[:, 4:])


# ------------------------------------------------------------------ whole model
@pytest.fixture(scope='module')
def hip_model(dev, golden):
    from yolov4_amd.yolo.model.yolov4 import YOLOv4
    m = YOLOv4(CFG, device=dev)
    sd = m.state_dict()
    recipe.fill_state_dict_(sd, int(golden('model')['seed']))
    m.load_state_dict(sd)
    return m.to(dev)


def _reset(m, golden):
    sd = m.state_dict()
    cpu = {k: v.cpu() for k, v in sd.items()}
    recipe.fill_state_dict_(cpu, int(golden('model')['seed']))
    m.load_state_dict(cpu)


def test_model_eval_golden(dev, golden, hip_model):
    from yolov4_amd.yolo.util.utils import postprocess
    g = golden('model')
    m = hip_model
    _reset(m, golden)
    recipe.calibrate_bn_(m, recipe.randn((8, 3, 64, 64), 77).to(dev))
    sd = m.state_dict()
    for k in g.files:
        if k.startswith('cal.'):
            close(sd[k[4:]], g[k], 1e-4, 1e-3)
    m.eval()
    with torch.no_grad():
        out = m(recipe.randn((2, 3, 64, 64), 78).to(dev))
    ref = g['eval64.out']
    close(out[..., 4:], ref[..., 4:], 1e-4, 1e-4, scale=False)     # obj / cls: abs 1e-4
    close(out[..., :4], ref[..., :4], 1e-4, 1e-4)                 # boxes in px: 1e-4 of the box scale
    det = postprocess(out.clone(), 80, 0.12, 0.4)
    for b in range(2):
        rd = g[f'eval64.det{b}']
        assert det[b].shape == rd.shape
        gd = det[b].cpu().numpy()
        assert np.array_equal(gd[:, 6], rd[:, 6])                  # same classes, same count per class
        # scores differ in the 6th digit, so two near-tied boxes of one class may swap places:
        # compare per class as sets (rows sorted by x1)
        for c in np.unique(rd[:, 6]):
            a = gd[gd[:, 6] == c]; r = rd[rd[:, 6] == c]
            close(a[np.argsort(a[:, 0])], r[np.argsort(r[:, 0])], 1e-4, 1e-4)
    _reset(m, golden)
    recipe.calibrate_bn_(m, recipe.randn((4, 3, 128, 128), 76).to(dev))
    m.eval()
    with torch.no_grad():
        out = m(recipe.randn((1, 3, 128, 128), 79).to(dev))
    ref = g['eval128.out']
    close(out[..., 4:], ref[..., 4:], 1e-4, 1e-4, scale=False)
    close(out[..., :4], ref[..., :4], 1e-4, 1e-4)


def test_model_train_step_golden(dev, golden, hip_model):
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    g = golden('model')
    m = hip_model
    _reset(m, golden)
    m.train()
    m.zero_grad(set_to_none=True)
    x = recipe.randn((2, 3, 128, 128), 80).to(dev)
    labels = recipe.synth_labels(2, 128, 81, counts=[9, 21])
    crit = YOLOLoss(CFG, ignore_thresh=0.7, device=dev, mutate_outputs=False)
    outs = m(x)
    for l in range(3):
        close(outs[l]['output'], g[f'train128.output{l}'], 1e-4, 1e-4, scale=False)
        close(outs[l]['pred'], g[f'train128.pred{l}'], 1e-4, 1e-4)
    loss = crit(outs, {'padded_labels': labels})
    ref = float(g['train128.loss'])
    assert abs(float(loss) - ref) <= 1e-4 * ref
    loss.backward()
    named = dict(m.named_parameters())
    # Gradients through 110 layers with B=2 batch statistics are ill-conditioned: the reference's own
    # fp32 CPU arithmetic is a few percent away from an fp64 evaluation of the same graph (see
    # test_gradients_within_reference_rounding).  Against the fp32 golden only a loose bound is meaningful.
    for kk, refn in zip([str(q) for q in g['train128.gradnorm_keys']], g['train128.gradnorm']):
        got = float(named[kk].grad.double().norm())
        assert abs(got - refn) <= 5e-2 * max(refn, 1e-6), (kk, got, refn)
    # (every stored train128.grad.* / gradslice.* tensor is compared under a conditioning-aware per-tensor bound in
    # tests/test_gpu_round2.py::test_train_step_stored_gradients_within_reference_rounding)
    for k in g.files:
        if k.startswith('train128.grad.head.'):            # the head is 2 layers from the loss: tight
            close(named[k[14:]].grad, g[k], 3e-4, 1e-3)
        if k.startswith('train128.after.'):
            close(m.state_dict()[k[15:]], g[k], 1e-5, 1e-4)


def test_wgrad_on_side_stream_is_identical(dev, golden, hip_model):
    """Opt-in mode: filter gradients produced on a second HIP stream and accumulated into .grad outside
    autograd.  Kernels are deterministic, so both schedules must give the same bits -- twice (accumulate)."""
    from yolov4_amd import ops
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    m = hip_model
    _reset(m, golden)
    m.train()
    x = recipe.randn((2, 3, 128, 128), 80).to(dev)
    labels = recipe.synth_labels(2, 128, 81, counts=[9, 21])
    crit = YOLOLoss(CFG, ignore_thresh=0.7, device=dev, mutate_outputs=False)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    grads = {}
    for on in (False, True):
        m.load_state_dict(sd)
        m.zero_grad(set_to_none=True)
        ops.set_async_wgrad(on)
        try:
            for _ in range(2):
                crit(m(x), {'padded_labels': labels}).backward()
        finally:
            ops.set_async_wgrad(False)
        grads[on] = {k: p.grad.clone() for k, p in m.named_parameters()}
    for k in grads[False]:
        assert torch.equal(grads[False][k], grads[True][k]), k


def test_ddp_grad_slots_written_in_place(dev, golden, hip_model):
    """BucketedDDP (single process, no process group): filter gradients are written by the wgrad kernel straight
    into the flat bucket slots; result must equal plain autograd accumulation, also over a 2-step window."""
    from yolov4_amd.ddp import BucketedDDP
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    m = hip_model
    _reset(m, golden)
    m.train()
    x = recipe.randn((2, 3, 128, 128), 80).to(dev)
    labels = recipe.synth_labels(2, 128, 81, counts=[9, 21])
    crit = YOLOLoss(CFG, ignore_thresh=0.7, device=dev, mutate_outputs=False)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.zero_grad(set_to_none=True)
    for _ in range(2):
        crit(m(x), {'padded_labels': labels}).backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.load_state_dict(sd)
    ddp = BucketedDDP(m)
    try:
        ddp.zero_grad()
        ddp.accumulating = True
        crit(ddp(x), {'padded_labels': labels}).backward()
        ddp.finish_backward()
        ddp.accumulating = False
        ddp.rearm()
        crit(ddp(x), {'padded_labels': labels}).backward()
        ddp.finish_backward()
        for k, p in m.named_parameters():
            assert p.grad.data_ptr() >= 0 and torch.equal(p.grad, ref[k]), k
        assert all(b.pending == 0 for b in ddp.buckets)
    finally:
        for h in ddp._hooks:
            h.remove()
        for p in m.parameters():
            p.grad = None
            for a in ('_y4_grad_fresh', '_y4_grad_ready', '_y4_ddp'):
                if hasattr(p, a):
                    delattr(p, a)


def test_gradients_within_reference_rounding(dev, golden, hip_model):
    """Same linear functional of the head logits on three backends: CPU fp64 (truth), CPU fp32 (the
    reference's arithmetic: torch ATen), HIP fp32.  The HIP path must be as close to the truth as the
    reference's own fp32 path is: per parameter within a factor 3 (+1e-5) of the reference's fp32
    error (individual ratios scatter), and no worse in the median (factor 1.5)."""
    m = hip_model
    _reset(m, golden)
    S, B = 128, 2
    sd = NW.empty_state_dict()
    recipe.fill_state_dict_(sd, int(golden('model')['seed']))
    x = recipe.randn((B, 3, S, S), 80)
    G = [recipe.randn((B, 255, S // s, S // s), 900 + i) for i, s in enumerate((8, 16, 32))]

    def run_cpu(dtype):
        net = NW.RefNet({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}, CFG)
        lg = net.forward_train(x.to(dtype))
        torch.autograd.backward(lg, [t.to(dtype) for t in G])
        return {k: v.grad.double() for k, v in net.p.items() if v.grad is not None}

    g64, g32 = run_cpu(torch.float64), run_cpu(torch.float32)
    m.train()
    m.zero_grad(set_to_none=True)
    lg = m.head.logits(*m.neck(*m.backbone(x.to(dev))))
    torch.autograd.backward(lg, [t.to(dev) for t in G])
    e32, eh = [], []
    for k, p in m.named_parameters():
        n = float(g64[k].norm())
        e32.append(float((g32[k] - g64[k]).norm()) / n)
        eh.append(float((p.grad.double().cpu() - g64[k]).norm()) / n)
        assert eh[-1] <= 3.0 * e32[-1] + 1e-5, (k, eh[-1], e32[-1])
    assert np.median(eh) <= 1.5 * np.median(e32) + 1e-6


def test_model_matches_oracle_at_608(dev, golden, hip_model):
    """BASELINE resolution (608x608, B=1), train-mode statistics, against the oracle run on
    the host here and now (no fixture: 22743 x 85 outputs)."""
    m = hip_model
    _reset(m, golden)
    x = recipe.randn((1, 3, 608, 608), 123)
    sd = NW.empty_state_dict()
    recipe.fill_state_dict_(sd, int(golden('model')['seed']))
    net = NW.RefNet(sd, CFG)
    with torch.no_grad():
        ref = net.forward_train(x)
    m.train()
    with torch.no_grad():
        p1, p2, p3 = m.neck(*m.backbone(x.to(dev)))
        lg = m.head.logits(p1, p2, p3)
    for a, b in zip(lg, ref):
        close(a, b, 2e-4, 1e-3)


def _iou_xyxy(b, others):
    tl = np.maximum(b[None, :2], others[:, :2]); br = np.minimum(b[None, 2:4], others[:, 2:4])
    inter = np.prod(np.clip(br - tl, 0, None), 1) * (tl < br).all(1)
    return inter / (np.prod(b[2:4] - b[:2]) + np.prod(others[:, 2:4] - others[:, :2], 1) - inter)


def _assert_difference_is_a_near_tie(a, r, pred, conf, nms_thre, tol=2e-4):
    """a, r: two [n,7] survivor lists that should be equal; pred [N,85] (xyxy already).  Every row present in only one of
    them must owe that to a decision within `tol` of its threshold: its score within tol of conf, or its IoU with some
    same-class candidate of higher score within tol of nms_thre (suppressed in one run, kept in the other)."""
    def key(row):
        return (int(row[6]), round(float(row[0]), 1), round(float(row[1]), 1))
    ka, kr = {key(x): x for x in a}, {key(x): x for x in r}
    odd = [ka[k] for k in ka.keys() - kr.keys()] + [kr[k] for k in kr.keys() - ka.keys()]
    assert 0 < len(odd) <= 4, len(odd)
    for row in odd:
        c = int(row[6])
        score = row[4] * row[5]
        if abs(score - conf) <= tol * max(conf, 1e-6):
            continue
        sc = pred[:, 4] * pred[:, 5 + c]
        cand = pred[(sc >= conf) & (sc > score)][:, :4]
        assert len(cand), row
        iou = _iou_xyxy(row[:4], cand)
        assert np.abs(iou - nms_thre).min() <= tol, (row, float(np.abs(iou - nms_thre).min()))


def test_eval_batch_independence_at_config2_size(dev, golden, hip_model):
    """BASELINE configs[1] size (608x608, bs=32, eval): every image of the batch must come out as it does alone.
    Size-independent property at the full configuration (the oracle needs ~1 s per image on the host; the single-image
    results are pinned to it by test_model_eval_golden / test_model_matches_oracle_at_608).  Tile shapes and the
    1x1 kernel choice depend on the number of rows, so equality is to rounding, not to the bit; NMS sets are exact."""
    from yolov4_amd.yolo.util.utils import postprocess
    m = hip_model
    _reset(m, golden)
    # random weights are only well-conditioned with BatchNorm statistics taken at the resolution in use
    recipe.calibrate_bn_(m, recipe.randn((8, 3, 608, 608), 77).to(dev))
    m.eval()
    x = recipe.randn((32, 3, 608, 608), 321).to(dev)
    with torch.no_grad():
        full = m(x)
        assert full.shape == (32, 22743, 85) and torch.isfinite(full).all()
        for i in (0, 13, 31):
            one = m(x[i:i + 1])
            close(full[i, :, 4:], one[0, :, 4:], 2e-5, 1e-4, scale=False)
            close(full[i, :, :4], one[0, :, :4], 2e-5, 1e-4)
            sc = (one[0, :, 4:5] * one[0, :, 5:]).flatten()
            thr = float(torch.sort(sc, descending=True).values[300])
            # keep clear of candidates whose score sits within rounding of the threshold
            near = ((sc - thr).abs() < 1e-5 * max(thr, 1e-6)).sum()
            if int(near) <= 1:
                da = postprocess(full[i:i + 1].clone(), 80, thr, 0.45)[0]
                db = postprocess(one.clone(), 80, thr, 0.45)[0]
                assert (da is None) == (db is None)
                if da is not None and da.shape != db.shape:
                    # the two runs differ by rounding (2e-5, checked above): a survivor set may differ ONLY where a
                    # decision sat within that rounding of its threshold -- proven here for every odd row out
                    _assert_difference_is_a_near_tie(da.cpu().numpy(), db.cpu().numpy(), one[0].cpu().numpy(), thr, 0.45)
                    continue
                if da is not None:
                    assert torch.equal(da[:, 6], db[:, 6])                   # same classes, same count per class
                    # scores agree to rounding only, so two near-tied boxes of one class may swap places in the
                    # score-ordered output: compare each class as a set (rows sorted by x1)
                    a, r = da.cpu().numpy(), db.cpu().numpy()
                    for c in np.unique(r[:, 6]):
                        aa, rr = a[a[:, 6] == c], r[r[:, 6] == c]
                        close(aa[np.argsort(aa[:, 0])][:, :6], rr[np.argsort(rr[:, 0])][:, :6], 2e-5, 1e-4)
