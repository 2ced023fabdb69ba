# -*- coding: utf-8 -*-
"""Host-side operators over libyolov4_amd.so.

PyTorch is used here for device memory (caching allocator), streams and the
autograd tape only; every arithmetic kernel on the path is a HIP kernel of the
library, reached through the C ABI (yolov4_amd/_lib.py).

Tensor convention: activations are logical NCHW torch tensors whose memory is
NHWC (torch.channels_last), possibly a channel slice of a wider NHWC buffer
(pixel pitch `ld` > C).  Conv weights are logical OIHW (the reference's
state_dict layout, darknet/darknet.py:31-36) stored channels_last == KRSC.
"""
import ctypes
import functools
import os

import torch

from ._lib import ACT_IDS, Y4Error, check, float_array, int_array, lib

CL = torch.channels_last


# ------------------------------------------------------------------ plumbing
def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _require_gpu(t, what):
    if not t.is_cuda:
        raise Y4Error(f'{what}: tensor is on {t.device}; the yolov4_amd hot path runs on an MI355X only '
                      '(no CPU fallback)')
    if t.dtype != torch.float32:
        raise Y4Error(f'{what}: expected float32, got {t.dtype}')


def nhwc_pitch(t):
    """Pixel pitch of a logical-NCHW tensor stored NHWC (possibly a channel
    slice), or None if the memory is not in that form."""
    if t.dim() != 4:
        return None
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if C > 1 and sc != 1:
        return None
    ld = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else C))
    if ld < C:
        return None
    if W > 1 and sw != ld:
        return None
    if H > 1 and sh != W * ld:
        return None
    if B > 1 and sb != H * W * ld:
        return None
    return ld


def empty_nhwc(B, C, H, W, device, pad_to=1):
    """[B,C,H,W] logical view over an NHWC buffer whose pitch is C rounded up to pad_to."""
    Cp = (C + pad_to - 1) // pad_to * pad_to
    buf = torch.empty((B, Cp, H, W), device=device, dtype=torch.float32, memory_format=CL)
    return buf if Cp == C else buf[:, :C]


def as_nhwc(t, need_vec4=True, min_pitch=0):
    """Return (tensor, ld) with NHWC memory, 16-B aligned base, ld % 4 == 0 and
    ld >= min_pitch; repacks (one torch copy, boundary plumbing) only when the
    caller handed a tensor in another layout."""
    ld = nhwc_pitch(t)
    ok = ld is not None and (not need_vec4 or (ld % 4 == 0 and t.data_ptr() % 16 == 0)) and ld >= min_pitch
    if ok:
        return t, ld
    B, C, H, W = t.shape
    pad = 1
    if need_vec4:
        pad = 4
    if min_pitch > C:
        pad = max(pad, 32)
    out = empty_nhwc(B, C, H, W, t.device, pad_to=pad)
    out.copy_(t)
    return out, nhwc_pitch(out)


def krsc(w):
    """Conv weight [Cout,Cin,k,k] as a KRSC device pointer holder (channels_last memory)."""
    if w.dim() != 4:
        raise Y4Error('conv weight must be 4-D')
    Co, Ci, kh, kw = w.shape
    exp = (kh * kw * Ci, 1, kw * Ci, Ci)
    st = w.stride()
    same = all(w.shape[i] == 1 or st[i] == exp[i] for i in range(4))
    return w if same else w.contiguous(memory_format=CL)


def _ws(nbytes, device):
    return torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------ operand maxima (conv mode 3, "f16x2")
# The two-piece fp16 split scales every conv operand by a power of two taken from max|tensor| (include/yolov4_amd.h).
# The maximum is a by-product of the kernel that PRODUCES the operand (BN+act forward, BN backward, conv epilogue):
# a device word ("cell") that travels with the tensor as the Python attribute `y4_amax`.  Cells of tensors that share
# a concat buffer are one shared cell (atomicMax).  A tensor without a cell is always legal: the consumer then spends
# one extra pass over it (ops.amax_raw / inside the library).
_AMAX = {'pool': {}, 'N': 8192}


def f16x2_mode():
    return lib().y4_get_conv_mode() == 3


def planes_mode():
    """Which form pre-split operands take in the current conv mode: 'f16x2' (mode 3: two fp16 pieces per element under a
    per-tensor scale), 'bf16' (mode 2, BASELINE configs[4]: plain bf16 values in the first half of each fp32-sized pixel row),
    None (the other modes have no DMA-fed kernels)."""
    m = lib().y4_get_conv_mode()
    if m == 2 or (m == 3 and lib().y4_get_planes_bf16()):
        return 'bf16'                                # (mode 3 + the switch: only the plane layers compute in bf16)
    return 'f16x2' if m == 3 else None


def new_amax(device, n=1):
    """n (1 or 8) zeroed device words out of a ring of 8192; each half is re-zeroed (one tiny fill) when the ring enters
    it, i.e. >= 4096 words (several training steps) after its cells were handed out.  Every cell carries the generation
    of its half: `live()` tells a holder whether the words are still its own."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _AMAX['pool'].get(key)
    if st is None:
        st = {'buf': torch.zeros(_AMAX['N'], dtype=torch.int32, device=device), 'i': 0, 'gen': [0, 0]}
        _AMAX['pool'][key] = st
    half = _AMAX['N'] // 2
    i = (st['i'] + n - 1) // n * n                    # n-word blocks are n-aligned, so they never straddle a half
    if i >= _AMAX['N']:
        i = 0
    if i // half != (st['i'] - 1) // half or st['i'] == 0:
        h0 = i // half * half
        st['buf'][h0:h0 + half].zero_()
        st['gen'][i // half] += 1                     # cells handed out of this half before now are dead
    st['i'] = i + n
    cell = st['buf'][i:i + n]
    cell.y4_gen = (st, i // half, st['gen'][i // half])
    return cell


# Pre-split operands for the DMA-fed conv kernels (csrc/conv_planes.hip); Y4_PLANES=0 keeps every tensor fp32 (A/B runs)
PLANES = {'on': os.environ.get('Y4_PLANES', '1') != '0'}
_PCELLS = {}


def planes_cell(device, n=8):
    """n zeroed device words for a PLANE tensor's scale.  Unlike ring cells these are never recycled: a pre-split tensor
    cannot be measured again, so its cell must live exactly as long as the tensor (the views keep their block alive)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _PCELLS.get(key)
    if st is None or st['i'] + n > st['buf'].numel():
        st = {'buf': torch.zeros(2048, dtype=torch.int32, device=device), 'i': 0}
        _PCELLS[key] = st
    cell = st['buf'][st['i']:st['i'] + n]
    st['i'] += n
    return cell


def live(cell):
    """cell if its words still belong to it, else None (its ring half was recycled: a tensor held across ~4096 cell
    allocations -- cached features, many forwards before one backward).  The holder then takes the maximum again
    (amax_raw / inside the library) instead of reading another tensor's.  Cells from elsewhere (amax_raw) never expire."""
    if cell is None:
        return None
    g = getattr(cell, 'y4_gen', None)
    if g is None:
        return cell
    st, half, gen = g
    return cell if st['gen'][half] == gen else None


def amax_of(t):
    return live(getattr(t, 'y4_amax', None)) if t is not None else None


def tag_amax(t, cell):
    if cell is not None and t is not None:
        t.y4_amax = cell
    return t


def amax_raw(t, valid_channels=None):
    """max|finite element| of an NHWC activation (one read pass) into a fresh cell."""
    B, C, H, W = t.shape
    t, ld = as_nhwc(t, need_vec4=False)
    cell = torch.empty(1, dtype=torch.int32, device=t.device)
    check(lib().y4_amax_f32(_ptr(t), ld, B * H * W, int(valid_channels or C), _ptr(cell), _stream()), 'amax')
    return cell


def amax_merge(dst, src):
    check(lib().y4_amax_merge_u32(_ptr(dst), _ptr(src), _stream()), 'amax_merge')


def conv_out_hw(H, W, k, s):
    p = (k - 1) // 2
    return (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1


# ------------------------------------------------------------------ raw op wrappers (no autograd)
def prepared_filter(param):
    """Inference only (conv mode 3): the filter's maximum + fp16 planes live in a buffer kept on the parameter and are
    REFRESHED on every call by y4_conv2d_prepare_filter_f32, which re-splits only when the filter's bits changed (exact
    checksum taken on the device: `.data` writes, raw-pointer optimizer kernels and load_state_dict are all seen)."""
    w = krsc(param)
    L = lib()
    Cout, K = w.shape[0], w.shape[1] * w.shape[2] * w.shape[3]
    key = (w.data_ptr(), Cout, K, w.device)
    hit = getattr(param, '_y4_prepared', None)
    if hit is None or hit[0] != key:
        nbytes = L.y4_conv2d_prepared_bytes(Cout, K)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        buf[:64].zero_()                              # header: "nothing prepared yet"
        hit = (key, buf)
        param._y4_prepared = hit
    buf = hit[1]
    check(L.y4_conv2d_prepare_filter_f32(_ptr(w), Cout, K, _ptr(buf), buf.numel(), _stream()), 'conv2d_prepare_filter')
    return buf


def fast_conv_shape(Cin, k, s):
    """Shapes the implicit-GEMM kernels take (everything YOLOv4 builds); anything else ConvBNAct accepts -- any channel
    count, odd kernel size, stride -- runs on the direct kernels of csrc/conv_generic.hip."""
    return k in (1, 3) and s in (1, 2) and (Cin % 32 == 0 or (Cin == 3 and k == 3 and s == 1))


def planes_stride2_ok(k, s, H, W):
    """The 3x3 stride-2 layers run on the plane kernels (forward; wgrad with x read at 4 p - 2 w) on even maps, f16x2 only."""
    # (bf16 operands: only in the hybrid mode, whose register-staged stride-2 dgrad is the f16x2 one; mode 2 keeps stride 2 off planes)
    return (s == 2 and k == 3 and H % 2 == 0 and W % 2 == 0 and _S2_PLANES
            and (planes_mode() == 'f16x2' or (planes_mode() == 'bf16' and f16x2_mode())))


@functools.lru_cache(maxsize=4096)
def _planes_fit(mode, B, H, W, ci, co, k, s, dgrad):
    return bool(lib().y4_conv_planes_fit(B, H, W, ci, co, k, s, 1 if dgrad else 0))


def planes_fit(B, H, W, ci, co, k, s, dgrad=True):
    """Can the DMA kernels address every operand of this layer (32-bit buffer windows, y4_conv_planes_fit)?  Asked BEFORE a
    producer is told to write its result pre-split, and before backward picks the plane dgrad of a stride-2 layer: a tensor
    that exists only as planes has no register-staged kernel to fall back to."""
    return _planes_fit((lib().y4_get_conv_mode(), lib().y4_get_planes_bf16()), int(B), int(H), int(W), int(ci), int(co),
                       int(k), int(s), bool(dgrad))


def s2_plane_dgrad(B, H, W, ci, co):
    """Does the dgrad of a 3x3 stride-2 plane layer (x: [B, ci, H, W]) run on the plane kernel (four parity-class launches over a
    pre-split dy)?  Not with fewer than 128 input channels (half-empty column tiles: 64->128 @304 took 1.81 ms there, 1.24 + 0.14
    on the register-staged kernel over an fp32 twin of dy), nor when the whole dx tensor is beyond the kernel's one 32-bit
    window.  Forward (which prepares the filter planes that dgrad will want) and backward ask the same question."""
    # (bf16 operands: from 64 input channels on -- one MFMA per product makes the half-empty tiles cheap: 670.1 / 676.4 -> 679.5 / 679.7 img/s)
    floor = 64 if planes_mode() == 'bf16' else 128
    return _S2_DGRAD_PLANES and ci >= floor and planes_fit(B, H, W, ci, co, 3, 2, dgrad=True)


_S2_PLANES = os.environ.get('Y4_PLANES_S2', '1') != '0'      # (A/B switch)
_S2_DGRAD_PLANES = os.environ.get('Y4_PLANES_S2_DGRAD', '1') != '0'      # (A/B switch: 0 = register-staged stride-2 dgrad over an fp32 twin of dy)
# bf16 conv RESULTS on the plane layers of conv mode 'bf16' (y4_conv2d_fwd_planes_f32 y_bf16; what autocast does to a conv's
# output): the three BatchNorm sweeps read y at half the width.  ON by default since round 4's rewrite of the sweeps: +2.8 %
# (618.4 / 619.5 vs 602.6 / 601.6 img/s at bs = 128, A/B on one box).  With the old sweeps it measured +-0 -- their run-time
# `ybf` branch made the backward sweeps wait for every bf16 load before issuing the next (pointwise.hip, "How the three
# BatchNorm sweeps address memory"), not, as first concluded, a limit on bytes in flight.  Y4_BF16_Y=0: fp32 results.
_BF16_Y = os.environ.get('Y4_BF16_Y', '1') != '0'


def conv_fwd_raw(x, w, k, s, scale=None, shift=None, act='linear', residual=None, out=None, out_pad=1, x_amax=None,
                 out_amax=None, w_prepared=None):
    L = lib()
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    Ho, Wo = conv_out_hw(H, W, k, s)
    if out is None:
        out = empty_nhwc(B, Cout, Ho, Wo, x.device, pad_to=out_pad)
    ldy = nhwc_pitch(out)
    w = krsc(w)
    if not fast_conv_shape(Cin, k, s) or (Cin == 3 and (Cout > 32 or residual is not None)):
        if k % 2 == 0:
            raise Y4Error('even kernel sizes are not supported (pad = (k - 1) // 2 would not keep the map size)')
        x, ldx = as_nhwc(x, need_vec4=False)
        ldr = 0
        if residual is not None:
            residual, ldr = as_nhwc(residual, need_vec4=False)
        check(L.y4_conv2d_generic_fwd_f32(_ptr(x), ldx, _ptr(w), _ptr(out), ldy, B, H, W, Cin, Cout, k, s, _ptr(scale), _ptr(shift),
                                          ACT_IDS[act], _ptr(residual), ldr, _stream()), 'conv2d_generic_fwd')
        return out
    if Cin == 3:
        if k != 3 or s != 1 or residual is not None:
            raise Y4Error('Cin=3 is supported for the 3x3/s1 stem only')
        sb, sc, sh, sw = x.stride()
        check(L.y4_conv2d_stem_fwd_f32(_ptr(x), sb, sc, sh, sw, _ptr(w), _ptr(out), ldy, B, H, W, Cout,
                                       _ptr(scale), _ptr(shift), ACT_IDS[act], None, _stream()), 'conv2d_stem_fwd')
        return out
    x, ldx = as_nhwc(x)
    ldr = 0
    if residual is not None:
        residual, ldr = as_nhwc(residual, need_vec4=False)
    if w_prepared is not None:
        check(L.y4_conv2d_fwd_prepared_f32(_ptr(x), ldx, _ptr(w_prepared), _ptr(out), ldy, B, H, W, Cin, Cout, k, s,
                                           _ptr(scale), _ptr(shift), ACT_IDS[act], _ptr(residual), ldr, _ptr(x_amax),
                                           _ptr(out_amax), _stream()), 'conv2d_fwd_prepared')
        return out
    nbytes = L.y4_conv2d_fwd_workspace(Cin, Cout, k)
    ws = _ws(nbytes, x.device)                  # the call's own filter planes: nothing is shared between calls
    check(L.y4_conv2d_fwd_f32(_ptr(x), ldx, _ptr(w), _ptr(out), ldy, B, H, W, Cin, Cout, k, s,
                              _ptr(scale), _ptr(shift), ACT_IDS[act], _ptr(residual), ldr, _ptr(x_amax), _ptr(out_amax),
                              _ptr(ws), nbytes, _stream()), 'conv2d_fwd')
    return out


def dgrad_filter_buffer(Cin, Cout, k, device):
    """Buffer a training forward call fills with the transposed filter planes for the backward pass of the same layer (one
    split launch serves both; the dgrad call takes it as its workspace and launches no filter kernels)."""
    return _ws(lib().y4_conv2d_dgrad_workspace(Cin, Cout, k), device)


def conv_fwd_bnstats_raw(x, w, k, s, running_mean, running_var, nbt, momentum, eps, x_amax=None, dgrad_filter=None):
    """Training-mode conv: raw output y + BatchNorm batch statistics taken in the conv epilogue
    (per-M-tile column sums, folded in fp64 by a second-stage kernel).  Returns (y, mean, invstd)."""
    L = lib()
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    Ho, Wo = conv_out_hw(H, W, k, s)
    if not fast_conv_shape(Cin, k, s) or (Cin == 3 and Cout > 32) or Cout % 4:
        # direct conv, statistics by a sweep over its output (y4_bn_stats_f32 needs 16-B rows: pitch rounded up to 4)
        if Cout % 4:
            raise Y4Error('BatchNorm over a channel count that is not a multiple of 4 is not supported')
        y = conv_fwd_raw(x, w, k, s, out=empty_nhwc(B, Cout, Ho, Wo, x.device, pad_to=4), x_amax=x_amax)
        mean, invstd = bn_stats_raw(y, running_mean, running_var, nbt, momentum, eps)
        return y, mean, invstd
    y = empty_nhwc(B, Cout, Ho, Wo, x.device)
    ldy = nhwc_pitch(y)
    w = krsc(w)
    pbytes = L.y4_conv2d_bnstats_workspace(B, H, W, Cin, Cout, k, s)
    part = _ws(pbytes, x.device)
    if Cin == 3:
        if k != 3 or s != 1:
            raise Y4Error('Cin=3 is supported for the 3x3/s1 stem only')
        sb, sc, sh, sw = x.stride()
        check(L.y4_conv2d_stem_fwd_f32(_ptr(x), sb, sc, sh, sw, _ptr(w), _ptr(y), ldy, B, H, W, Cout,
                                       None, None, ACT_IDS['linear'], _ptr(part), _stream()), 'conv2d_stem_fwd')
        nparts = (B * H * W + 255) // 256
    else:
        x, ldx = as_nhwc(x)
        n = ctypes.c_longlong(0)
        nbytes = L.y4_conv2d_fwd_workspace(Cin, Cout, k)
        cws = _ws(nbytes, x.device)
        check(L.y4_conv2d_fwd_bnstats_f32(_ptr(x), ldx, _ptr(w), _ptr(y), ldy, B, H, W, Cin, Cout, k, s,
                                          _ptr(part), pbytes, ctypes.byref(n), _ptr(x_amax), _ptr(cws), nbytes,
                                          _ptr(dgrad_filter), dgrad_filter.numel() if dgrad_filter is not None else 0, _stream()),
              'conv2d_fwd_bnstats')
        nparts = n.value
    mean = torch.empty(Cout, device=x.device, dtype=torch.float32)
    invstd = torch.empty(Cout, device=x.device, dtype=torch.float32)
    wsb = L.y4_bn_finalize_workspace(Cout)
    ws = _ws(wsb, x.device)
    check(L.y4_bn_finalize_partials_f32(_ptr(part), nparts, B * Ho * Wo, Cout, _ptr(mean), _ptr(invstd),
                                        _ptr(running_mean), _ptr(running_var), _ptr(nbt), float(momentum), float(eps),
                                        _ptr(ws), wsb, _stream()), 'bn_finalize_partials')
    return y, mean, invstd


# ------------------------------------------------------------------ pre-split operands ("planes", csrc/conv_planes.hip)
class Planes:
    """An NHWC activation stored as the two fp16 pieces of the f16x2 split: per pixel and 32-channel K tile
    [64 B hi | 64 B lo] (4 bytes per element, like fp32), with the device word that fixes its power-of-two scale."""
    __slots__ = ('buf', 'shape', 'amax')

    def __init__(self, buf, shape, amax):
        self.buf, self.shape, self.amax = buf, tuple(shape), amax


class PlanesTensor(torch.Tensor):
    """What a pre-split activation looks like from Python: float32-TYPED (autograd only differentiates floating outputs, and
    the bytes travel through autograd.Function like any activation) but its 4 bytes per element are two fp16 pieces, not a
    float.  Only ConvBNAct consumes it (ops.planes_of); everything an observer would do with an activation -- .cpu(),
    .numpy(), arithmetic, indexing, printing values, torch.save -- fails loudly instead of yielding garbage.  A module with
    forward hooks never hands one out (darknet.observed): its result is then fp32 with the pre-split copy as `y4_twin`."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    def _refuse(self, *a, **kw):
        raise Y4Error('this tensor holds a PRE-SPLIT activation (two fp16 pieces per element, csrc/conv_planes.hip), not '
                      'float32 values: only ConvBNAct can consume it.  Register the forward hook before the forward pass '
                      '(hooked modules return fp32), or run with Y4_PLANES=0 to keep every intermediate tensor fp32.')

    cpu = to = numpy = tolist = item = float = double = half = bfloat16 = int = long = clone = _refuse
    sum = mean = abs = max = min = amax = amin = norm = std = var = exp = log = sigmoid = tanh = relu = _refuse
    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __truediv__ = __rtruediv__ = __neg__ = __pow__ = _refuse
    __iadd__ = __isub__ = __imul__ = __itruediv__ = __matmul__ = __getitem__ = __setitem__ = _refuse
    __lt__ = __le__ = __gt__ = __ge__ = __float__ = __int__ = __bool__ = __array__ = __dlpack__ = _refuse
    __reduce_ex__ = __reduce__ = __deepcopy__ = _refuse                      # torch.save, pickle, copy.deepcopy

    def __repr__(self, *a, **kw):
        return (f'PlanesTensor(shape={tuple(self.shape)}, device={self.device}: pre-split fp16 pairs, NOT float32 values; '
                'see yolov4_amd.ops.PlanesTensor)')

    __str__ = __repr__

    def __format__(self, spec):
        return self.__repr__()

    def detach(self):
        return as_planes(torch.Tensor.detach(self), getattr(self, 'y4_amax', None))


def as_planes(t, cell=None):
    """Mark tensor t (same memory, still attached to the autograd graph) as a pre-split activation."""
    if type(t) is not PlanesTensor:
        t = t.as_subclass(PlanesTensor)
    t.y4_planes = True
    if cell is not None:
        t.y4_amax = cell
    return t


def planes_split_raw(x, amax=None):
    """fp32 NHWC tensor -> Planes (one read + one write pass; tensors whose producer does not emit planes itself)."""
    L = lib()
    B, C, H, W = x.shape
    x, ldx = as_nhwc(x)
    if planes_mode() == 'bf16':
        amax = None                                  # plain bf16 values: no scale
    else:
        if amax is None:
            amax = amax_of(x)
        if amax is None:
            amax = amax_raw(x)
    buf = torch.empty((B, H, W, C * 4), dtype=torch.uint8, device=x.device)
    check(L.y4_planes_split_f32(_ptr(x), ldx, B * H * W, C, _ptr(amax), _ptr(buf), _stream()), 'planes_split')
    return Planes(buf, (B, C, H, W), amax)


def conv_fwd_planes_raw(xp, w, k, s, stats=True, dgrad_filter=None, y_bf16=False, bias=None, out=None):
    """Training-mode conv over a Planes input: raw output y (+ per-M-tile column sums, n_tiles).  y_bf16: y leaves as plain
    bf16 in the first half of each row of the (float32-typed) result -- an internal tensor only the BatchNorm sweeps read.
    bias (with stats=False): added to every result row (a conv without BatchNorm); out: destination of y's shape."""
    L = lib()
    B, Cin, H, W = xp.shape
    Cout = w.shape[0]
    Ho, Wo = conv_out_hw(H, W, k, s)
    y = out if _slot_ok(out, (B, Cout, Ho, Wo)) else empty_nhwc(B, Cout, Ho, Wo, xp.buf.device)
    w = krsc(w)
    nbytes = L.y4_conv2d_fwd_workspace(Cin, Cout, k)
    ws = _ws(nbytes, xp.buf.device)
    pbytes = ((B * Ho * Wo + 127) // 128) * 2 * Cout * 4 if stats else 0       # 128-row tiles at most
    part = _ws(pbytes, xp.buf.device) if stats else None
    n = ctypes.c_longlong(0)
    check(L.y4_conv2d_fwd_planes_f32(_ptr(xp.buf), _ptr(w), _ptr(y), nhwc_pitch(y), B, H, W, Cin, Cout, k, s,
                                     _ptr(part), pbytes, ctypes.byref(n), _ptr(xp.amax), _ptr(ws), nbytes,
                                     _ptr(dgrad_filter), dgrad_filter.numel() if dgrad_filter is not None else 0,
                                     1 if y_bf16 else 0, _ptr(bias), _stream()),
          'conv2d_fwd_planes')
    return (y, part, n.value) if stats else y


def conv_fwd_planes_bnstats_raw(xp, w, k, s, running_mean, running_var, nbt, momentum, eps, dgrad_filter=None, y_bf16=False):
    """conv_fwd_bnstats_raw over a Planes input: (y, mean, invstd)."""
    L = lib()
    y, part, nparts = conv_fwd_planes_raw(xp, w, k, s, dgrad_filter=dgrad_filter, y_bf16=y_bf16)
    Cout = w.shape[0]
    M = y.shape[0] * y.shape[2] * y.shape[3]
    mean = torch.empty(Cout, device=y.device, dtype=torch.float32)
    invstd = torch.empty(Cout, device=y.device, dtype=torch.float32)
    wsb = L.y4_bn_finalize_workspace(Cout)
    ws = _ws(wsb, y.device)
    check(L.y4_bn_finalize_partials_f32(_ptr(part), nparts, M, Cout, _ptr(mean), _ptr(invstd),
                                        _ptr(running_mean), _ptr(running_var), _ptr(nbt), float(momentum), float(eps),
                                        _ptr(ws), wsb, _stream()), 'bn_finalize_partials')
    return y, mean, invstd


def planes_of(t):
    """Planes view of a tensor written pre-split by bn_act_fwd_raw / bn_act_bwd_raw (tag y4_planes); None otherwise."""
    if t is None or not getattr(t, 'y4_planes', False):
        return None
    return Planes(t, t.shape, getattr(t, 'y4_amax', None))


def conv_dgrad_planes_raw(dyp, w, x_shape, k, residual=None, prepared=None, s=1):
    """dx of a conv from a Planes dy: stride 1 (the DMA forward kernel on the mirrored transposed filter), or 3x3 stride 2 on an
    even map (one launch per parity class of dx)."""
    L = lib()
    B, Cin, H, W = x_shape
    Cout = w.shape[0]
    dx = empty_nhwc(B, Cin, H, W, dyp.buf.device)
    nbytes = L.y4_conv2d_dgrad_workspace(Cin, Cout, k)
    ws = prepared if prepared is not None else _ws(nbytes, dyp.buf.device)     # prepared: filled by the forward call
    ldr = 0
    if residual is not None:
        residual, ldr = as_nhwc(residual)
    check(L.y4_conv2d_dgrad_planes_f32(_ptr(dyp.buf), None if prepared is not None else _ptr(krsc(w)), _ptr(dx), nhwc_pitch(dx), B, H, W, Cin, Cout, k,
                                       s, _ptr(ws), nbytes, _ptr(dyp.amax), _ptr(residual), ldr, _stream()), 'conv2d_dgrad_planes')
    return dx


def conv_wgrad_planes_raw(xp, dyp, w_shape, k, out=None, s=1):
    """dW of a conv from Planes x and dy (stride 1, or the 3x3 stride-2 layers on even maps); `out` as conv_wgrad_raw."""
    L = lib()
    B, Cin, H, W = xp.shape
    Cout = w_shape[0]
    if out is not None and tuple(out.shape) == tuple(w_shape) and _is_krsc_dense(out) and out.data_ptr() % 16 == 0:
        dw = out
        WGRAD_STATS['in_place'] += 1
    else:
        WGRAD_STATS['temporary'] += 1
        dw = krsc(torch.empty(w_shape, device=xp.buf.device, dtype=torch.float32).contiguous(memory_format=CL))
    nbytes = L.y4_conv2d_wgrad_planes_workspace(B, H, W, Cin, Cout, k, s)
    ws = _ws(nbytes, xp.buf.device)
    check(L.y4_conv2d_wgrad_planes_f32(_ptr(xp.buf), _ptr(dyp.buf), _ptr(dw), B, H, W, Cin, Cout, k, s, _ptr(ws), nbytes,
                                       _ptr(xp.amax), _ptr(dyp.amax), _stream()), 'conv2d_wgrad_planes')
    return dw


def conv_stem_dgrad_raw(dy, w, x):
    """Gradient wrt the 3-channel network input (same shape and strides as x)."""
    L = lib()
    B, _, H, W = x.shape
    dy, lddy = as_nhwc(dy, need_vec4=False)
    dx = torch.empty_like(x, memory_format=torch.preserve_format)
    sb, sc, sh, sw = dx.stride()
    check(L.y4_conv2d_stem_dgrad_f32(_ptr(dy), lddy, _ptr(krsc(w)), _ptr(dx), sb, sc, sh, sw, B, H, W, w.shape[0], _stream()),
          'conv2d_stem_dgrad')
    return dx


def last_conv_kernel():
    """Symbol (as rocprofv3 prints it) of the conv kernel this thread launched last, '' if unknown; measurement aid."""
    import ctypes
    buf = ctypes.create_string_buffer(160)
    check(lib().y4_last_conv_kernel(buf, 160), 'last_conv_kernel')
    return buf.value.decode()


def conv_dgrad_raw(dy, w, x_shape, k, s, dy_amax=None, residual=None, prepared=None):
    L = lib()
    B, Cin, H, W = x_shape
    Cout = w.shape[0]
    if k not in (1, 3) or s not in (1, 2):
        dy, lddy = as_nhwc(dy, need_vec4=False)
        dx = empty_nhwc(B, Cin, H, W, dy.device)
        ldr = 0
        if residual is not None:
            residual, ldr = as_nhwc(residual, need_vec4=False)
        check(L.y4_conv2d_generic_dgrad_f32(_ptr(dy), lddy, _ptr(krsc(w)), _ptr(dx), nhwc_pitch(dx), B, H, W, Cin, Cout, k, s,
                                            _ptr(residual), ldr, _stream()), 'conv2d_generic_dgrad')
        return dx
    cpad = (Cout + 31) // 32 * 32
    dy, lddy = as_nhwc(dy, min_pitch=cpad)
    dx = empty_nhwc(B, Cin, H, W, dy.device)
    nbytes = L.y4_conv2d_dgrad_workspace(Cin, Cout, k)
    ws = prepared if prepared is not None else _ws(nbytes, dy.device)          # prepared: filled by the forward call (mode 3)
    ldr = 0
    if residual is not None:
        residual, ldr = as_nhwc(residual, need_vec4=False)
    check(L.y4_conv2d_dgrad_f32(_ptr(dy), lddy, None if prepared is not None else _ptr(krsc(w)), _ptr(dx), nhwc_pitch(dx), B, H, W, Cin, Cout, k, s,
                                _ptr(ws), nbytes, _ptr(dy_amax), _ptr(residual), ldr, _stream()),
          'conv2d_dgrad')
    return dx


def _is_krsc_dense(t):
    """Dense KRSC memory; strides of size-1 dims are irrelevant (a 1x1 filter slot of a flat gradient bucket is a
    plain [Co, Ci, 1, 1] view whose kh/kw strides read 1)."""
    Co, Ci, kh, kw = t.shape
    exp = (kh * kw * Ci, 1, kw * Ci, Ci)
    st = t.stride()
    return t.dtype == torch.float32 and all(t.shape[i] == 1 or st[i] == exp[i] for i in range(4))


WGRAD_STATS = {'in_place': 0, 'temporary': 0}



def conv_wgrad_raw(x, dy, w_shape, k, s, out=None, x_amax=None, dy_amax=None):
    """out: optional fp32 tensor of shape w_shape whose memory is dense KRSC (e.g. a gradient slot of a flat
    DDP bucket): the kernel then writes the filter gradient in place."""
    L = lib()
    B, Cin, H, W = x.shape
    Cout = w_shape[0]
    if out is not None and tuple(out.shape) == tuple(w_shape) and _is_krsc_dense(out) and out.data_ptr() % 16 == 0:
        dw = out
        WGRAD_STATS['in_place'] += 1
    else:
        WGRAD_STATS['temporary'] += 1
        dw = torch.empty(w_shape, device=dy.device, dtype=torch.float32).contiguous(memory_format=CL)
        dw = krsc(dw)
    if k not in (1, 3) or s not in (1, 2) or (Cin % 4 and Cin != 3) or (Cin == 3 and (k != 3 or s != 1 or Cout > 32)):
        xs, ldx = as_nhwc(x, need_vec4=False)
        dys, lddy = as_nhwc(dy, need_vec4=False)
        check(L.y4_conv2d_generic_wgrad_f32(_ptr(xs), ldx, _ptr(dys), lddy, _ptr(dw), B, H, W, Cin, Cout, k, s, _stream()),
              'conv2d_generic_wgrad')
        return dw
    if Cin == 3:
        dy, lddy = as_nhwc(dy, need_vec4=False)
        nbytes = L.y4_conv2d_stem_wgrad_workspace(B, H, W, Cout)
        ws = _ws(nbytes, dy.device)
        sb, sc, sh, sw = x.stride()
        check(L.y4_conv2d_stem_wgrad_f32(_ptr(x), sb, sc, sh, sw, _ptr(dy), lddy, _ptr(dw), B, H, W, Cout,
                                         _ptr(ws), nbytes, _stream()), 'conv2d_stem_wgrad')
        return dw
    x, ldx = as_nhwc(x)
    dy, lddy = as_nhwc(dy, min_pitch=(Cout + 3) // 4 * 4)
    nbytes = L.y4_conv2d_wgrad_workspace(B, H, W, Cin, Cout, k, s)
    ws = _ws(nbytes, dy.device)
    check(L.y4_conv2d_wgrad_f32(_ptr(x), ldx, _ptr(dy), lddy, _ptr(dw), B, H, W, Cin, Cout, k, s,
                                _ptr(ws), nbytes, _ptr(x_amax), _ptr(dy_amax), _stream()), 'conv2d_wgrad')
    return dw


def bn_stats_raw(y, running_mean, running_var, nbt, momentum, eps):
    L = lib()
    B, C, H, W = y.shape
    y, ld = as_nhwc(y)
    mean = torch.empty(C, device=y.device, dtype=torch.float32)
    invstd = torch.empty(C, device=y.device, dtype=torch.float32)
    nbytes = L.y4_bn_workspace(B * H * W, C)
    ws = _ws(nbytes, y.device)
    check(L.y4_bn_stats_f32(_ptr(y), ld, B * H * W, C, _ptr(mean), _ptr(invstd), _ptr(running_mean),
                            _ptr(running_var), _ptr(nbt), float(momentum), float(eps), _ptr(ws), nbytes, _stream()),
          'bn_stats')
    return mean, invstd


def _slot_ok(out, shape):
    """out: a caller-provided destination (typically a channel slice of a concat buffer)."""
    if out is None:
        return False
    ld = nhwc_pitch(out)
    return (tuple(out.shape) == tuple(shape) and out.dtype == torch.float32 and ld is not None and ld % 4 == 0
            and out.data_ptr() % 16 == 0)


def bn_act_fwd_raw(y, mean, invstd, gamma, beta, act, residual=None, out=None, out_amax=None, planes=False, y_bf16=False, cell=None):
    """planes: z leaves PRE-SPLIT for the plane conv kernels (a float32-typed tensor whose 4 bytes per element hold the two
    fp16 pieces; tagged y4_planes), scaled by an analytic bound of max|z| that the call leaves in the tensor's cell (no
    measuring pass; with a residual whose maximum is unknown: one measure-only launch first)."""
    L = lib()
    B, C, H, W = y.shape
    y, ldy = as_nhwc(y)
    ldr = 0
    if residual is not None:
        residual, ldr = as_nhwc(residual)
    yb = 16 if y_bf16 else 0                         # y holds bf16 values (conv_fwd_planes_raw y_bf16)
    if planes == 'slot':
        # out: a slot of a pre-split CatBuffer (its cell already holds the joint bound of all producers: scale as given)
        cb = out.y4_cat
        if residual is not None or not _slot_ok(out, (B, C, H, W)):
            raise Y4Error('a pre-split concat slot takes a BatchNorm result of its own shape, without a skip operand')
        bf = planes_mode() == 'bf16'
        zp = out.data_ptr()
        if bf:
            # bf16 rows keep their values in the FIRST HALF of the fp32-sized pixel row: channel c of the concat sits at byte 2 c of
            # the row, not at the slot's fp32 address 4 c
            zp = cb.buf.data_ptr() + (zp - cb.buf.data_ptr()) // 2
        check(L.y4_bn_act_fwd_f32(_ptr(y), ldy, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), ACT_IDS[act], None, 0,
                                  ctypes.c_void_p(zp), nhwc_pitch(out), B * H * W, C, None if bf else _ptr(cb.cell[0:1]),
                                  (3 if bf else 1) + yb, None, None, _stream()), 'bn_act_fwd(planes slot)')
        return out
    if planes:
        both = planes == 'both'
        zp = empty_nhwc(B, C, H, W, y.device)        # the pre-split tensor (float32-typed, 4 bytes per element)
        z = (out if _slot_ok(out, (B, C, H, W)) else empty_nhwc(B, C, H, W, y.device)) if both else zp
        args = (_ptr(y), ldy, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), ACT_IDS[act], _ptr(residual), ldr)
        if planes_mode() == 'bf16':                  # conv mode 2: plain bf16 values in the first half of each row, no scale
            check(L.y4_bn_act_fwd_f32(*args, _ptr(z), nhwc_pitch(z), B * H * W, C, None, 3 + yb, None,
                                      _ptr(zp) if both else None, _stream()), 'bn_act_fwd(bf16)')
            zp = as_planes(zp)
            if both:
                z.y4_twin = zp
                return z
            return zp
        given = cell is not None                     # cell: the scale is ALREADY fixed (the joint bound of a pre-split concat buffer)
        if given and (residual is not None or both):
            raise Y4Error('a given plane scale goes with a plain planes-only result')
        cell = cell if given else planes_cell(y.device)
        res_cell = amax_of(residual) if residual is not None else None
        if given:
            mode = 1
        elif residual is not None and res_cell is None:
            check(L.y4_bn_act_fwd_f32(*args, None, C, B * H * W, C, _ptr(cell), yb, None, None, _stream()), 'bn_act_fwd(measure)')
            mode = 1
        else:
            mode = 2
        check(L.y4_bn_act_fwd_f32(*args, _ptr(z), nhwc_pitch(z), B * H * W, C, _ptr(cell), mode + yb, _ptr(res_cell),
                                  _ptr(zp) if both else None, _stream()), 'bn_act_fwd(planes)')
        zp = as_planes(zp, cell)
        if both:
            z.y4_twin = zp                           # fp32 for everybody else, planes for the conv that can take them
            return tag_amax(z, cell[0:1])
        return as_planes(zp, cell[0:1])
    z = out if _slot_ok(out, (B, C, H, W)) else empty_nhwc(B, C, H, W, y.device)
    check(L.y4_bn_act_fwd_f32(_ptr(y), ldy, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), ACT_IDS[act],
                              _ptr(residual), ldr, _ptr(z), nhwc_pitch(z), B * H * W, C, _ptr(out_amax), yb, None, None, _stream()),
          'bn_act_fwd')
    return z


def bn_act_bwd_raw(dz, y, mean, invstd, gamma, beta, act, dgamma_out=None, dbeta_out=None, out_amax=None, planes=None,
                   frozen=False, bf16=False, twin=False, y_bf16=False):
    """dgamma_out / dbeta_out: optional contiguous fp32 [C] destinations (gradient slots of a flat DDP bucket).
    twin (with planes): dy stays fp32 and the pre-split copy is returned beside it: (dy, dgamma, dbeta, dy_planes)."""
    L = lib()
    B, C, H, W = y.shape
    dz, lddz = as_nhwc(dz)
    y, ldy = as_nhwc(y)
    dy = empty_nhwc(B, C, H, W, y.device)

    def slot(t):
        ok = t is not None and t.dtype == torch.float32 and tuple(t.shape) == (C,) and t.is_contiguous() and t.is_cuda
        return t if ok else torch.empty(C, device=y.device, dtype=torch.float32)
    dgamma, dbeta = slot(dgamma_out), slot(dbeta_out)
    dyp = empty_nhwc(B, C, H, W, y.device) if twin else None
    nbytes = L.y4_bn_workspace(B * H * W, C)
    ws = _ws(nbytes, y.device)
    check(L.y4_bn_act_bwd_f32(_ptr(dz), lddz, _ptr(y), ldy, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta),
                              ACT_IDS[act], _ptr(dy), nhwc_pitch(dy), _ptr(dgamma), _ptr(dbeta), B * H * W, C,
                              _ptr(ws), nbytes, _ptr(out_amax), _ptr(planes), _ptr(dyp) if twin else None,
                              (1 if frozen else 0) | (2 if bf16 else 0) | (4 if y_bf16 else 0), _stream()),
          'bn_act_bwd')
    if twin:
        return dy, dgamma, dbeta, dyp
    return dy, dgamma, dbeta


def bias_grad_raw(dy):
    L = lib()
    B, C, H, W = dy.shape
    dy, ld = as_nhwc(dy, need_vec4=False)
    db = torch.empty(C, device=dy.device, dtype=torch.float32)
    nbytes = L.y4_bias_grad_workspace(B * H * W, C)
    ws = _ws(nbytes, dy.device)
    check(L.y4_bias_grad_f32(_ptr(dy), ld, B * H * W, C, _ptr(db), _ptr(ws), nbytes, _stream()), 'bias_grad')
    return db


def bn_fold_raw(gamma, beta, rm, rv, eps):
    L = lib()
    C = gamma.numel()
    scale = torch.empty(C, device=gamma.device, dtype=torch.float32)
    shift = torch.empty(C, device=gamma.device, dtype=torch.float32)
    check(L.y4_bn_fold_f32(_ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv), float(eps), _ptr(scale), _ptr(shift), C,
                           _stream()), 'bn_fold')
    return scale, shift


def add_raw(a, b):
    L = lib()
    B, C, H, W = a.shape
    a, lda = as_nhwc(a)
    b, ldb = as_nhwc(b)
    out = empty_nhwc(B, C, H, W, a.device)
    check(L.y4_add_f32(_ptr(a), lda, _ptr(b), ldb, _ptr(out), nhwc_pitch(out), B * H * W, C, _stream()), 'add')
    return out


def copy_into_raw(src, dst):
    """dst (a channel slice of an NHWC buffer) <- src"""
    L = lib()
    B, C, H, W = src.shape
    src, lds = as_nhwc(src)
    ldd = nhwc_pitch(dst)
    if ldd is None:
        raise Y4Error('copy_into: destination is not NHWC')
    check(L.y4_copy_channels_f32(_ptr(src), lds, _ptr(dst), ldd, B * H * W, C, _stream()), 'copy_channels')


# ------------------------------------------------------------------ wgrad on a side stream
# The filter gradient of a layer is off the critical path of backward (only the optimizer / the gradient
# exchange consume it), while the next thing on the path -- the BatchNorm backward sweeps of the layer below --
# is HBM-bound and leaves the matrix cores idle.  Issuing every wgrad on a second HIP stream lets the hardware
# co-schedule the two kinds of work; the main stream joins the side stream once, at the end of backward.
# Measured on MI355X (bs=64 @608, A/B on one box): 274.6/275.6 img/s off vs 274.2/279.8 on -- inside the
# run-to-run noise, so it is opt-in (Y4_ASYNC_WGRAD=1 or set_async_wgrad(True)), covered by a parity test.
_ASYNC = {'on': os.environ.get('Y4_ASYNC_WGRAD', '0') == '1', 'streams': {}, 'join_queued': False}


def set_async_wgrad(on):
    _ASYNC['on'] = bool(on)


def side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _ASYNC['streams'].get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _ASYNC['streams'][key] = st
    return st


def join_side_stream(device=None):
    """Make the current stream wait for all side-stream work issued so far (end of backward)."""
    for st in _ASYNC['streams'].values():
        torch.cuda.current_stream(st.device).wait_stream(st)
    _ASYNC['join_queued'] = False


def _wgrad_to_param(x, dy, param, k, s, x_amax=None, dy_amax=None, fn=None):
    """wgrad on the side stream, accumulated straight into param.grad (autograd gets None for this input).  fn: the wgrad
    call to make there (default: the register-staged kernel on fp32 operands)."""
    main = torch.cuda.current_stream(x.device)
    side = side_stream(x.device)
    ev = main.record_event()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        dw = fn() if fn is not None else conv_wgrad_raw(x, dy, tuple(param.shape), k, s, x_amax=x_amax, dy_amax=dy_amax)
        if param.grad is None:
            dw.record_stream(main)          # allocated in the side stream's pool, consumed (and freed) on the main one
            param.grad = dw
        else:
            param.grad.record_stream(side)
            param.grad.add_(dw)
        ready = getattr(param, '_y4_grad_ready', None)
        if ready is not None:
            ready()
    x.record_stream(side)
    dy.record_stream(side)
    for cell in (x_amax, dy_amax):                   # operand maxima are read by the side-stream kernel as well
        if cell is not None:
            cell.record_stream(side)
    if not _ASYNC['join_queued']:
        _ASYNC['join_queued'] = True
        torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)


def _pad_out_channels(weight, bias, cop):
    """Filter [Cout, Cin, k, k] (+ bias) with zero rows up to cop output channels (whole K tiles for the plane dgrad / wgrad)."""
    co = weight.shape[0]
    if cop == co:
        return weight, bias
    wp = torch.zeros((cop,) + tuple(weight.shape[1:]), dtype=weight.dtype, device=weight.device).contiguous(memory_format=torch.channels_last)
    wp[:co].copy_(weight)
    bp = None
    if bias is not None:
        bp = torch.zeros(cop, dtype=bias.dtype, device=bias.device)
        bp[:co].copy_(bias)
    return wp, bp


def _split_padded(t, cop, amax=None):
    """Planes [B, cop, H, W] of the NHWC gradient t [B, C, H, W], C <= cop: channels C .. cop - 1 leave as zeros.  Reads the pad
    of t's rows where its pixel pitch covers cop channels (the head gradients leave yolo_decode_bwd with pitch 256) -- never
    writes it: the memory behind a gradient's logical channels need not be ours -- else goes through a padded copy."""
    B, C, H, W = t.shape
    t, ld = as_nhwc(t)
    if ld < cop:
        v = empty_nhwc(B, cop, H, W, t.device)
        v[:, :C].copy_(t)
        t, ld = v, nhwc_pitch(v)
    buf = torch.empty((B, H, W, cop * 4), dtype=torch.uint8, device=t.device)
    check(lib().y4_planes_split_into_f32(_ptr(t), ld, B * H * W, cop, _ptr(amax), _ptr(buf), cop, C, _stream()), 'planes_split(padded)')
    return Planes(buf, (B, cop, H, W), amax)


# ------------------------------------------------------------------ autograd functions
class ConvBNActFn(torch.autograd.Function):
    """conv -> BatchNorm -> activation (+ skip), darknet/darknet.py:53-58 (+ :76-80)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, residual, cfg):
        k, s, act, training, bn = cfg['k'], cfg['s'], cfg['act'], cfg['training'], cfg['bn']
        _require_gpu(x, 'ConvBNAct input')
        _require_gpu(weight, 'ConvBNAct weight')
        # The destination slot is handed in through cfg but must NOT stay reachable from ctx: forward returns that
        # very tensor, so output -> grad_fn -> ctx -> cfg['out'] -> output would be a reference cycle that only a
        # full gc.collect() breaks (round 1: +4.96 GiB of concat buffers per training step).  ctx keeps the
        # scalars and parameter handles backward needs, nothing else.
        dest = cfg.pop('out', None)
        ctx.cfg = {key: cfg.get(key) for key in ('k', 's', 'act', 'gamma_param', 'beta_param', 'weight_param',
                                                 'dres_put', 'dres_take', 'dx_put')}
        ctx.x_shape = tuple(x.shape)
        ctx.has_res = residual is not None
        # conv mode 3: operand maxima travel with the tensors (see "operand maxima" above)
        f16 = f16x2_mode() and x.shape[1] != 3
        bfm = planes_mode() == 'bf16'                # conv mode 2: plane layers run the DMA kernels on bf16 operands
        # pre-split operands (conv_planes.hip): x arrives as planes when its producer was asked to (out_planes, below);
        # all three of this layer's convs then run on the DMA kernels, dy leaving the BatchNorm backward as planes too
        xp = planes_of(x)
        if xp is None and cfg.get('x_twin') is not None:
            xp = planes_of(cfg['x_twin'])            # x itself is fp32 (it has other consumers); its pre-split twin feeds this conv
        ctx.x_planes = xp is not None
        nobn_planes = xp is not None and not bn and act == 'linear' and residual is None and s == 1 and cfg.get('grad', True)
        if xp is not None and not ((f16 or bfm) and x.shape[1] % 32 == 0 and (nobn_planes or (
                bn and training and weight.shape[0] % 32 == 0 and (s == 1 or planes_stride2_ok(k, s, x.shape[2], x.shape[3]))))):
            raise Y4Error('a pre-split (planes) tensor reached a conv that cannot consume it')
        x_amax = live(cfg.get('x_amax')) if f16 else None
        z_amax = None
        io = cfg.get('io')
        if xp is not None:
            x_amax = xp.amax
        elif f16 and x_amax is None and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            x_amax = amax_raw(x)                     # needed twice (forward, wgrad): one pass here instead of two inside
        ctx.x_amax = x_amax
        if bn and training:
            if x.shape[0] * conv_out_hw(x.shape[2], x.shape[3], k, s)[0] * conv_out_hw(x.shape[2], x.shape[3], k, s)[1] <= 1:
                raise ValueError('Expected more than 1 value per channel when training')   # as nn.BatchNorm2d
            # the backward pass wants the transposed planes of the same filter: the forward split launch writes both
            ctx.dgrad_filter = None
            if ((f16 or (bfm and xp is not None)) and ctx.needs_input_grad[0] and x.shape[1] % 32 == 0 and fast_conv_shape(x.shape[1], k, s)
                    and weight.shape[0] % 4 == 0 and (xp is None or weight.shape[0] % 32 == 0)
                    # (bf16 planes at stride 2 whose dgrad stays on the register-staged f16x2 kernel: that one splits the filter itself)
                    and not (bfm and xp is not None and s == 2
                             and not s2_plane_dgrad(x.shape[0], x.shape[2], x.shape[3], x.shape[1], weight.shape[0]))):
                ctx.dgrad_filter = dgrad_filter_buffer(x.shape[1], weight.shape[0], k, x.device)
            # conv mode 'bf16': the plane conv writes y as bf16 (half the bytes for the three BatchNorm sweeps that read it)
            ybf = ctx.y_bf16 = bool(bfm and xp is not None and weight.shape[0] % 32 == 0 and _BF16_Y)
            if xp is not None:
                y, mean, invstd = conv_fwd_planes_bnstats_raw(xp, weight, k, s, cfg['running_mean'], cfg['running_var'],
                                                              cfg['nbt'], cfg['momentum'], cfg['eps'], dgrad_filter=ctx.dgrad_filter,
                                                              y_bf16=ybf)
            else:
                y, mean, invstd = conv_fwd_bnstats_raw(x, weight, k, s, cfg['running_mean'], cfg['running_var'],
                                                       cfg['nbt'], cfg['momentum'], cfg['eps'], x_amax=x_amax,
                                                       dgrad_filter=ctx.dgrad_filter)
            want = cfg.get('out_planes')
            cb = cfg.get('out_cat')
            if cb is not None and cb.planes:
                # a slot of a pre-split concat buffer (CatBuffer planes_norms): the activation goes there as planes
                z = bn_act_fwd_raw(y, mean, invstd, gamma, beta, act, residual, out=dest, planes='slot', y_bf16=ybf)
                if io is not None:
                    io['z_planes'] = True
                z_amax = cb.cell[0:1] if cb.cell is not None else None
            elif want and (f16 or bfm) and y.shape[1] % 32 == 0 and (want == 'both' or dest is None):
                sf = cfg.get('scale_from')               # a pre-split CatBuffer this result will be copied into (Upsample)
                z = bn_act_fwd_raw(y, mean, invstd, gamma, beta, act, residual, out=dest if want == 'both' else None,
                                   planes='both' if want == 'both' else True, y_bf16=ybf,
                                   cell=sf.cell if (sf is not None and sf.planes and sf.cell is not None and want is True) else None)
                z_amax = getattr(z, 'y4_amax', None)
                if io is not None:
                    io['z_planes'] = want != 'both'
                    io['z_twin'] = getattr(z, 'y4_twin', None)
            else:
                if f16x2_mode():
                    z_amax = live(cfg.get('out_amax')) if (dest is not None and _slot_ok(dest, tuple(y.shape))) else None
                    if z_amax is None:
                        z_amax = new_amax(x.device)
                z = bn_act_fwd_raw(y, mean, invstd, gamma, beta, act, residual, out=dest, out_amax=z_amax, y_bf16=ybf)
            ctx.save_for_backward(xp.buf if xp is not None else x, weight, y, mean, invstd, gamma, beta)
            ctx.mode = 'bn_train'
        elif bn and cfg.get('grad', True) and any(ctx.needs_input_grad[i] for i in (0, 1, 3, 4)):
            # eval-mode BatchNorm UNDER autograd (frozen statistics, e.g. fine-tuning): the unfused sequence, so that
            # backward has the pre-BN tensor; running statistics play mean / invstd and receive no batch terms
            y = conv_fwd_raw(x, weight, k, s, x_amax=x_amax, out=empty_nhwc(x.shape[0], weight.shape[0],
                                                                              *conv_out_hw(x.shape[2], x.shape[3], k, s), x.device, pad_to=4))
            invstd, _ = bn_fold_raw(torch.ones_like(gamma), beta, cfg['running_mean'], cfg['running_var'], cfg['eps'])
            mean = cfg['running_mean']
            if f16x2_mode():
                z_amax = live(cfg.get('out_amax')) if (dest is not None and _slot_ok(dest, tuple(y.shape))) else None
                if z_amax is None:
                    z_amax = new_amax(x.device)
            z = bn_act_fwd_raw(y, mean, invstd, gamma, beta, act, residual, out=dest, out_amax=z_amax)
            ctx.save_for_backward(x, weight, y, mean.detach().clone(), invstd, gamma, beta)
            ctx.mode = 'bn_eval_grad'
        elif bn:
            # no autograd in flight (val.py / detect.py): the filter planes and the BN fold are per-parameter-version caches
            frozen = not cfg.get('grad', True) and os.environ.get('Y4_NO_INFER_CACHE') != '1'
            # shapes outside the implicit-GEMM kernels run on the direct kernels (conv_generic.hip): those take fp32 filters
            # as they are (no prepared planes: K need not be a multiple of 32) and leave no maximum behind, so their result
            # stays untagged and the consumer measures it itself
            fast = f16 and fast_conv_shape(x.shape[1], k, s)
            wprep = prepared_filter(cfg['weight_param']) if (frozen and fast and cfg.get('weight_param') is not None) else None
            scale, shift = bn_fold_raw(gamma, beta, cfg['running_mean'], cfg['running_var'], cfg['eps'])
            o = dest
            Ho, Wo = conv_out_hw(x.shape[2], x.shape[3], k, s)
            o = o if _slot_ok(o, (x.shape[0], weight.shape[0], Ho, Wo)) else None
            if fast:
                z_amax = live(cfg.get('out_amax')) if o is not None else None
                if z_amax is None:                   # (never truth-test a device tensor: that is a host sync)
                    z_amax = new_amax(x.device)
            z = conv_fwd_raw(x, weight, k, s, scale, shift, act, residual, out=o, x_amax=x_amax, out_amax=z_amax,
                             w_prepared=wprep)
            ctx.mode = 'bn_eval'
        elif nobn_planes:
            # a conv WITHOUT BatchNorm over a pre-split input (the head's output convs): its Cout (255) is padded with zero
            # filters to whole K tiles, so that forward, dgrad and wgrad are plain plane-kernel calls; the bias rides the
            # forward kernel's skip-operand epilogue (one row for every pixel)
            co = weight.shape[0]
            q = 64 if bfm else 32
            cop = (co + q - 1) // q * q
            wpad, bpad = _pad_out_channels(weight, bias, cop)
            Ho, Wo = conv_out_hw(x.shape[2], x.shape[3], k, s)
            zf = empty_nhwc(x.shape[0], cop, Ho, Wo, x.device)
            ctx.dgrad_filter = dgrad_filter_buffer(x.shape[1], cop, k, x.device) if ctx.needs_input_grad[0] else None
            conv_fwd_planes_raw(xp, wpad, k, s, stats=False, dgrad_filter=ctx.dgrad_filter, bias=bpad, out=zf)
            z = zf[:, :co]
            ctx.save_for_backward(xp.buf, weight)
            ctx.mode = 'nobn_linear'
        else:
            wprep = prepared_filter(cfg['weight_param']) if (not cfg.get('grad', True) and f16 and fast_conv_shape(x.shape[1], k, s)
                                                              and os.environ.get('Y4_NO_INFER_CACHE') != '1'
                                                              and cfg.get('weight_param') is not None) else None
            z = conv_fwd_raw(x, weight, k, s, None, bias, act, residual, out_pad=32, x_amax=x_amax, w_prepared=wprep)
            if act != 'linear':
                ctx.mode = 'nobn_act'
            else:
                ctx.save_for_backward(x, weight)
                ctx.mode = 'nobn_linear'
        if io is not None:
            io['z_amax'] = z_amax
        return z

    @staticmethod
    def backward(ctx, dz):
        cfg = ctx.cfg
        k, s, act = cfg['k'], cfg['s'], cfg['act']
        f16 = f16x2_mode() and ctx.x_shape[1] != 3
        dy_amax = dy_pl = dyP = wq = None          # dyP / wq: pre-split dy and padded filter of a conv without BatchNorm over planes
        x_amax = live(ctx.x_amax)                     # None if the ring recycled it since forward: wgrad takes its own pass
        x_planes = getattr(ctx, 'x_planes', False)
        if ctx.mode in ('bn_train', 'bn_eval_grad'):
            x, weight, y, mean, invstd, gamma, beta = ctx.saved_tensors
            gp, bp = cfg.get('gamma_param'), cfg.get('beta_param')
            sink = (gp is not None and bp is not None and getattr(gp, '_y4_grad_fresh', False)
                    and getattr(bp, '_y4_grad_fresh', False) and gp.grad is not None and bp.grad is not None
                    and ctx.needs_input_grad[3] and ctx.needs_input_grad[4])
            # a layer whose input came pre-split runs dgrad and wgrad on the plane kernels: dy leaves the BatchNorm
            # backward sweep already split, scaled by a bound of max|dy| the reduce pass derives (word [5] of the cell)
            bfp = x_planes and planes_mode() == 'bf16'       # conv mode 2: dy leaves as plain bf16, no scale word
            planes = planes_cell(dz.device, 8) if (x_planes and not bfp) else None
            dy_amax = planes[5:6] if planes is not None else (new_amax(dz.device) if f16 else None)
            # stride 2 over planes: wgrad runs on the plane kernel; dgrad too, class by class (s2_plane_dgrad) -- or, where that
            # does not pay or does not fit, on the register-staged parity-class kernel, which wants fp32: dy then leaves the sweep
            # both ways (the bound in word [5] dominates max|dy|: it serves both as scale)
            twin_dy = x_planes and s == 2 and ctx.needs_input_grad[0] and not s2_plane_dgrad(
                ctx.x_shape[0], ctx.x_shape[2], ctx.x_shape[3], ctx.x_shape[1], dz.shape[1])
            res = bn_act_bwd_raw(dz, y, mean, invstd, gamma, beta, act,
                                 gp.grad if sink else None, bp.grad if sink else None,
                                 out_amax=None if planes is not None else dy_amax, planes=planes,
                                 frozen=ctx.mode == 'bn_eval_grad', bf16=bfp, twin=twin_dy, y_bf16=getattr(ctx, 'y_bf16', False))
            dy, dgamma, dbeta = res[:3]
            dy_pl = res[3] if twin_dy else dy
            if sink and dgamma is gp.grad and dbeta is bp.grad:
                # written straight into the (zeroed) DDP gradient slots: no temporaries, no accumulate kernels
                gp._y4_grad_fresh = bp._y4_grad_fresh = False
                gp._y4_grad_ready()
                bp._y4_grad_ready()
                dgamma = dbeta = None
            dbias = None
        elif ctx.mode == 'nobn_linear':
            x, weight = ctx.saved_tensors
            dy = dz
            dgamma = dbeta = None
            dbias = bias_grad_raw(dz) if ctx.needs_input_grad[2] else None
            if x_planes:
                # plane dgrad / wgrad want dy pre-split, over whole K tiles: the gradient with zero pad channels (a view of
                # what yolo_decode_bwd wrote), one measuring pass (f16x2) and one split pass; the filter padded alike
                bfp = planes_mode() == 'bf16'
                q = 64 if bfp else 32
                cop = (weight.shape[0] + q - 1) // q * q
                dyP = _split_padded(dz, cop, None if bfp else amax_raw(dz))
                wq = _pad_out_channels(weight, None, cop)[0]
            elif f16:
                dy_amax = amax_raw(dy)               # used by dgrad and wgrad: one pass instead of two
        else:
            raise Y4Error(f'backward through ConvBNAct in mode {ctx.mode} is not implemented')
        dx = None
        take = cfg.get('dres_take')
        skip_grad = take.pop('dres', None) if take is not None else None
        if take is not None and skip_grad is None:
            take['taker_done'] = True                # (a fork box whose other branch has not run yet: it must not park any more)
        if ctx.needs_input_grad[0]:
            if ctx.x_shape[1] == 3 and k == 3 and s == 1 and weight.shape[0] <= 32 and skip_grad is None:
                dx = conv_stem_dgrad_raw(dy, weight, x)       # gradient wrt the network input (never needed in training)
            # skip_grad: the gradient that reached this ResBlock unit over its skip connection, parked by the unit's
            # 3x3 conv (dres_put below): added in the dgrad epilogue instead of by a separate fan-in kernel
            elif x_planes and dyP is not None:
                dx = conv_dgrad_planes_raw(dyP, wq, ctx.x_shape, k, residual=skip_grad, prepared=getattr(ctx, 'dgrad_filter', None))
            elif x_planes and (s == 1 or not twin_dy):
                dx = conv_dgrad_planes_raw(Planes(dy, dy.shape, dy_amax), weight, ctx.x_shape, k, residual=skip_grad,
                                           prepared=getattr(ctx, 'dgrad_filter', None), s=s)
            else:
                dx = conv_dgrad_raw(dy, weight, ctx.x_shape, k, s, dy_amax=dy_amax, residual=skip_grad,
                                    prepared=getattr(ctx, 'dgrad_filter', None) if f16x2_mode() else None)
            ctx.dgrad_filter = None
        elif skip_grad is not None:
            raise Y4Error('a parked skip gradient has no consumer (input of the 1x1 conv does not require grad)')
        # dx_put: this conv and another one read the same tensor (a CSP fork); whichever backward runs first parks its dx here,
        # the other adds it in its dgrad epilogue (dres_take) -- the fork's fan-in add costs one operand read instead of a pass
        fbox = cfg.get('dx_put')
        if fbox is not None and dx is not None and not fbox.get('taker_done'):
            fbox['dres'] = dx
            dx = None
        dw = None
        if ctx.needs_input_grad[1]:
            param = cfg.get('weight_param')

            def wgrad(out=None):
                if x_planes and dyP is not None:     # (dy's rows hold whole K tiles, dW the rows of the real output channels)
                    return conv_wgrad_planes_raw(Planes(x, ctx.x_shape, ctx.x_amax), dyP, tuple(weight.shape), k, out=out)
                if x_planes:
                    return conv_wgrad_planes_raw(Planes(x, ctx.x_shape, ctx.x_amax), Planes(dy_pl, dy.shape, dy_amax),
                                                 tuple(weight.shape), k, out=out, s=s)
                return conv_wgrad_raw(x, dy, tuple(weight.shape), k, s, out=out, x_amax=x_amax, dy_amax=dy_amax)
            if _ASYNC['on'] and param is not None and param.requires_grad:
                # lands in param.grad on the side stream
                # (the operand the side stream reads: for a stride-2 plane layer that is the pre-split twin of dy)
                _wgrad_to_param(x, (dyP.buf if dyP is not None else dy_pl) if x_planes else dy, param, k, s,
                                ctx.x_amax if x_planes else x_amax, dyP.amax if dyP is not None else dy_amax,
                                fn=wgrad if x_planes else None)
            elif param is not None and getattr(param, '_y4_grad_fresh', False) and param.grad is not None:
                # gradient slot owned by BucketedDDP and still zero in this window: the kernel writes it in place
                # (no temporary, no accumulate pass); the bucket is told directly, autograd gets None
                param._y4_grad_fresh = False
                got = wgrad(out=param.grad)
                if got is not param.grad:
                    param.grad.add_(got)
                param._y4_grad_ready()
            else:
                dw = wgrad()
        dres = dz if ctx.has_res else None
        put = cfg.get('dres_put')
        if dres is not None and put is not None:
            put['dres'] = dres                      # consumed by the unit's 1x1 conv (dres_take); autograd sees no gradient
            dres = None
        return dx, dw, dbias, dgamma, dbeta, dres, None


class Fork2Fn(torch.autograd.Function):
    """A tensor consumed twice; the backward fan-in add is a library kernel
    instead of autograd's implicit accumulation."""

    @staticmethod
    def forward(ctx, x):
        # a branch without gradient must arrive as None, not as a materialised zero tensor (which would cost a fill,
        # a layout repack and an add): e.g. the skip branch of a ResBlock unit, whose gradient is folded into the
        # 1x1 conv's dgrad epilogue (ConvBNActFn dres_put / dres_take)
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None:
            return g2
        if g2 is None:
            return g1
        return add_raw(g1, g2)


def fork(x):
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    a, b = Fork2Fn.apply(x)
    cell = amax_of(x)
    twin = getattr(x, 'y4_twin', None)
    if twin is not None:
        a.y4_twin = b.y4_twin = twin
    if getattr(x, 'y4_planes', False):               # a pre-split tensor read by two plane-taking convs (CSP split convs)
        cell = getattr(x, 'y4_amax', None)
        return as_planes(a, cell), as_planes(b, cell)
    return tag_amax(a, cell), tag_amax(b, cell)


class AddFn(torch.autograd.Function):
    """a + b as a library kernel (the unfused ResBlock skip, used only where a hooked container forces the plain call path)."""

    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a, 'add input')
        _require_gpu(b, 'add input')
        return add_raw(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


class CatBuffer:
    """A concat destination allocated BEFORE its inputs are produced: `slot(i)` is the channel slice input i will
    occupy; a producer that accepts `out=` (ConvBNAct, Upsample) writes there directly and `cat(..., into=)` then
    has nothing to copy for it (inputs produced elsewhere are copied as usual)."""

    def __init__(self, B, sizes, H, W, device, planes_norms=None):
        self.sizes = list(sizes)
        self.buf = empty_nhwc(B, sum(self.sizes), H, W, device)
        self.offsets = [sum(self.sizes[:i]) for i in range(len(self.sizes))]
        # conv mode 3: ONE operand-maximum cell for the whole buffer, every producer folds into it
        self.amax = new_amax(device) if (device.type == 'cuda' and f16x2_mode()) else None
        # planes_norms (one training-mode BatchNorm2d per slot): the buffer is a PRE-SPLIT tensor -- every slot is written as
        # planes by its producer's BatchNorm sweep (the slot's channel tiles of the [pixel][C/32][128 B] rows), the concat is
        # consumed by a conv on the DMA kernels.  f16x2 planes share ONE scale: the joint analytic bound of all producers,
        # formed here, before any of them runs (y4_bn_planes_bound_f32 chained through its floor word).
        self.planes = False
        pm = planes_mode()
        q = 64 if pm == 'bf16' else 32
        if planes_norms is not None and pm is not None and device.type == 'cuda' and all(c % q == 0 for c in self.sizes):
            self.planes = True
            self.cell = None
            if pm == 'f16x2':
                self.cell = planes_cell(device)
                for i, n in enumerate(planes_norms):
                    if isinstance(n, torch.Tensor) and not isinstance(n, torch.nn.Module):
                        # an EXISTING fp32 tensor that cat() will split into its slot (PANBlock: the lateral input): its maximum
                        c = live(amax_of(n))
                        amax_merge(self.cell[0:1], c if c is not None else amax_raw(n))
                        continue
                    # (norm, M): a producer whose BatchNorm ran over M pixels of its own (the layer in front of an Upsample)
                    n, M = n if isinstance(n, tuple) else (n, B * H * W)
                    check(lib().y4_bn_planes_bound_f32(_ptr(n.weight), _ptr(n.bias), self.sizes[i], M,
                                                       _ptr(self.cell[0:1]) if i else None, _ptr(self.cell[0:1]), _stream()),
                          'bn_planes_bound')
                self.amax = self.cell[0:1]

    def slot(self, i):
        t = tag_amax(self.buf[:, self.offsets[i]:self.offsets[i] + self.sizes[i]], self.amax)
        if self.planes:
            t.y4_cat = self                          # ConvBNActFn writes its activation here pre-split (cfg 'out_cat')
        return t


def _split_into_slot(t, dst, cb):
    """fp32 NHWC tensor t -> the slot `dst` (a channel slice of cb.buf) of a pre-split CatBuffer, in the buffer's plane form."""
    B, C, H, W = t.shape
    t, ldt = as_nhwc(t)
    zp = dst.data_ptr()
    if planes_mode() == 'bf16':                      # bf16 rows: channel c of the concat sits at byte 2 c of the pixel row
        zp = cb.buf.data_ptr() + (zp - cb.buf.data_ptr()) // 2
    check(lib().y4_planes_split_into_f32(_ptr(t), ldt, B * H * W, C, _ptr(cb.cell[0:1]) if cb.cell is not None else None,
                                         ctypes.c_void_p(zp), nhwc_pitch(dst), 0, _stream()), 'planes_split_into')


class CatFn(torch.autograd.Function):
    """torch.cat(dim=1): copies into channel slices (skipped for inputs already produced in place, see
    CatBuffer); backward hands out slice views (zero copy)."""

    @staticmethod
    def forward(ctx, into, *xs):
        B, _, H, W = xs[0].shape
        ctx.sizes = [t.shape[1] for t in xs]
        if into is not None and (into.sizes != ctx.sizes or tuple(into.buf.shape) != (B, sum(ctx.sizes), H, W)):
            into = None
        out = into.buf if into is not None else empty_nhwc(B, sum(ctx.sizes), H, W, xs[0].device)
        o = 0
        for t in xs:
            _require_gpu(t, 'cat input')
            dst = out[:, o:o + t.shape[1]]
            if not (t.data_ptr() == dst.data_ptr() and t.stride() == dst.stride()):
                if into is not None and into.planes:
                    if getattr(t, 'y4_planes', False):
                        raise Y4Error('a pre-split concat buffer takes pre-split inputs only where their producers wrote them in place')
                    _split_into_slot(t, dst, into)   # an fp32 input: split (instead of copied) under the buffer's joint scale
                else:
                    copy_into_raw(t, dst)
            elif into is not None and into.planes != bool(getattr(t, 'y4_planes', False)):
                raise Y4Error('concat buffer and input disagree on the pre-split form')
            o += t.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        outs, o = [None], 0
        for c in ctx.sizes:
            outs.append(g[:, o:o + c])
            o += c
        return tuple(outs)


def cat(xs, into=None):
    out = CatFn.apply(into, *xs)
    if into is not None and into.planes and out.data_ptr() == into.buf.data_ptr():
        return as_planes(out, into.cell[0:1] if into.cell is not None else None)
    if into is not None and into.amax is not None and out.data_ptr() == into.buf.data_ptr():
        for t in xs:                                 # inputs that were copied in bring their own maximum
            cell = amax_of(t)
            if cell is None:
                cell = amax_raw(t)
            if cell.data_ptr() != into.amax.data_ptr():
                amax_merge(into.amax, cell)
        tag_amax(out, into.amax)
    return out


def cat_buffer(like, sizes, hw=None, planes_norms=None):
    """CatBuffer for inputs shaped like `like` ([B, *, H, W]; hw overrides the spatial size).  planes_norms: see CatBuffer."""
    H, W = hw if hw is not None else (like.shape[2], like.shape[3])
    return CatBuffer(like.shape[0], sizes, H, W, like.device, planes_norms=planes_norms)


class SppPoolCatFn(torch.autograd.Function):
    """SPPBlock.forward, yolo/model/yolov4.py:66-72: cat([pool5(x), pool9(x), pool5(x), x]) --
    max_pool3 (13) is constructed but never used."""

    @staticmethod
    def forward(ctx, x, cell=None):
        """cell: planes mode (spp_pool_cat planes=True) -- the result leaves PRE-SPLIT for a conv on the DMA kernels (one split pass
        over the concatenated tensor; f16x2: scaled by *cell, the maximum of x, which pooling cannot exceed)."""
        L = lib()
        _require_gpu(x, 'SPP input')
        B, C, H, W = x.shape
        x, ldx = as_nhwc(x)
        out = empty_nhwc(B, 4 * C, H, W, x.device)
        ldo = nhwc_pitch(out)
        need_idx = ctx.needs_input_grad[0]
        idx5 = torch.empty((B, H, W, C), dtype=torch.int8, device=x.device) if need_idx else None
        idx9 = torch.empty((B, H, W, C), dtype=torch.int8, device=x.device) if need_idx else None
        st = _stream()
        check(L.y4_maxpool_s1_fwd_f32(_ptr(x), ldx, _ptr(out[:, 0:C]), ldo, _ptr(idx5), B, H, W, C, 5, st), 'maxpool5')
        check(L.y4_maxpool_s1_fwd_f32(_ptr(x), ldx, _ptr(out[:, C:2 * C]), ldo, _ptr(idx9), B, H, W, C, 9, st), 'maxpool9')
        copy_into_raw(out[:, 0:C], out[:, 2 * C:3 * C])
        copy_into_raw(x, out[:, 3 * C:4 * C])
        ctx.idx = (idx5, idx9)
        ctx.shape = (B, C, H, W)
        if cell is not None:
            pl = empty_nhwc(B, 4 * C, H, W, x.device)
            check(L.y4_planes_split_f32(_ptr(out), ldo, B * H * W, 4 * C, _ptr(cell.t), _ptr(pl), st), 'planes_split(spp)')
            return pl
        return out

    @staticmethod
    def backward(ctx, g):
        L = lib()
        B, C, H, W = ctx.shape
        idx5, idx9 = ctx.idx
        g, ldg = as_nhwc(g)
        dx = empty_nhwc(B, C, H, W, g.device)
        ldx = nhwc_pitch(dx)
        st = _stream()
        copy_into_raw(g[:, 3 * C:4 * C], dx)
        for sl, idx, ks in ((g[:, 0:C], idx5, 5), (g[:, C:2 * C], idx9, 9), (g[:, 2 * C:3 * C], idx5, 5)):
            check(L.y4_maxpool_s1_bwd_f32(_ptr(sl), ldg, _ptr(idx), _ptr(dx), ldx, 1, B, H, W, C, ks, st), 'maxpool_bwd')
        return dx, None


def spp_pool_cat(x, planes=False):
    """cat([pool5(x), pool9(x), pool5(x), x]); max|out| = max|x| (pooling selects elements).  planes: the result leaves
    pre-split for its one consumer, a conv on the DMA kernels (one extra split pass; needs 4 C to be whole K tiles)."""
    pm = planes_mode()
    if planes and pm is not None and x.is_cuda and (4 * x.shape[1]) % (64 if pm == 'bf16' else 32) == 0:
        cell = None
        if pm == 'f16x2':
            # the scale word lives with the pre-split tensor (ring cells are recycled): a copy of x's maximum
            cell = planes_cell(x.device)
            src = live(amax_of(x))
            amax_merge(cell[0:1], src if src is not None else amax_raw(x))
        out = SppPoolCatFn.apply(x, Slot(cell[0:1]) if cell is not None else Slot(None))
        return as_planes(out, cell[0:1] if cell is not None else None)
    return tag_amax(SppPoolCatFn.apply(x), amax_of(x))


class MaxPoolS1Fn(torch.autograd.Function):
    """nn.MaxPool2d(k, stride=1, padding=k//2)."""

    @staticmethod
    def forward(ctx, x, ksize):
        L = lib()
        _require_gpu(x, 'maxpool input')
        B, C, H, W = x.shape
        x, ldx = as_nhwc(x)
        out = empty_nhwc(B, C, H, W, x.device)
        idx = torch.empty((B, H, W, C), dtype=torch.int8, device=x.device)
        check(L.y4_maxpool_s1_fwd_f32(_ptr(x), ldx, _ptr(out), nhwc_pitch(out), _ptr(idx), B, H, W, C, ksize,
                                      _stream()), 'maxpool')
        ctx.idx, ctx.ks, ctx.shape = idx, ksize, (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        L = lib()
        B, C, H, W = ctx.shape
        g, ldg = as_nhwc(g)
        dx = empty_nhwc(B, C, H, W, g.device)
        check(L.y4_maxpool_s1_bwd_f32(_ptr(g), ldg, _ptr(ctx.idx), _ptr(dx), nhwc_pitch(dx), 0, B, H, W, C, ctx.ks,
                                      _stream()), 'maxpool_bwd')
        return dx, None


class Upsample2xFn(torch.autograd.Function):
    """Upsample.forward, yolo/model/yolov4.py:82-90 (nearest, exact x2)."""

    @staticmethod
    def forward(ctx, x, out=None):
        L = lib()
        _require_gpu(x, 'upsample input')
        B, C, H, W = x.shape
        xpl = bool(getattr(x, 'y4_planes', False))
        cb = getattr(out.t, 'y4_cat', None) if out is not None else None
        x, ldx = as_nhwc(x)
        if xpl or (cb is not None and cb.planes):
            # a pre-split tensor into its slot of a pre-split concat buffer: pixel rows are copied as they are (f16x2: the
            # source was written under the buffer's joint scale; bf16: the C values sit in the first 2 C bytes of the row and
            # go to byte 2 * (channel offset) of the concat's row)
            if not (xpl and cb is not None and cb.planes and _slot_ok(out.t, (B, C, 2 * H, 2 * W))):
                raise Y4Error('upsample: a pre-split tensor goes into a pre-split concat slot of its shape, and nothing else does')
            dst, cols = out.t.data_ptr(), C
            if planes_mode() == 'bf16':
                dst, cols = cb.buf.data_ptr() + (dst - cb.buf.data_ptr()) // 2, C // 2
            check(L.y4_upsample2x_fwd_f32(_ptr(x), ldx, ctypes.c_void_p(dst), nhwc_pitch(out.t), B, H, W, cols, _stream()), 'upsample(planes)')
            ctx.shape = (B, C, H, W)
            return out.t
        out = out.t if (out is not None and _slot_ok(out.t, (B, C, 2 * H, 2 * W))) else empty_nhwc(B, C, 2 * H, 2 * W, x.device)
        check(L.y4_upsample2x_fwd_f32(_ptr(x), ldx, _ptr(out), nhwc_pitch(out), B, H, W, C, _stream()), 'upsample')
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        L = lib()
        B, C, H, W = ctx.shape
        g, ldg = as_nhwc(g)
        dx = empty_nhwc(B, C, H, W, g.device)
        check(L.y4_upsample2x_bwd_f32(_ptr(g), ldg, _ptr(dx), nhwc_pitch(dx), B, H, W, C, _stream()), 'upsample_bwd')
        return dx, None


class UpsampleNearestFn(torch.autograd.Function):
    """Upsample.forward for any target size, yolo/model/yolov4.py:82-90: the train branch is
    F.interpolate(size=target, mode='nearest'), the eval branch an integer-factor expand."""

    @staticmethod
    def forward(ctx, x, Ho, Wo, integer_factor, out=None):
        L = lib()
        _require_gpu(x, 'upsample input')
        B, C, H, W = x.shape
        x, ldx = as_nhwc(x)
        dst = out.t if (out is not None and _slot_ok(out.t, (B, C, Ho, Wo))) else empty_nhwc(B, C, Ho, Wo, x.device)
        check(L.y4_upsample_nearest_fwd_f32(_ptr(x), ldx, _ptr(dst), nhwc_pitch(dst), B, H, W, Ho, Wo, C,
                                            1 if integer_factor else 0, _stream()), 'upsample_nearest')
        ctx.meta = (B, C, H, W, Ho, Wo, integer_factor)
        return dst

    @staticmethod
    def backward(ctx, g):
        L = lib()
        B, C, H, W, Ho, Wo, integer_factor = ctx.meta
        g, ldg = as_nhwc(g)
        dx = empty_nhwc(B, C, H, W, g.device)
        check(L.y4_upsample_nearest_bwd_f32(_ptr(g), ldg, _ptr(dx), nhwc_pitch(dx), B, H, W, Ho, Wo, C,
                                            1 if integer_factor else 0, _stream()), 'upsample_nearest_bwd')
        return dx, None, None, None, None


def _dense(t):
    """t if its memory is one dense block (any dim order), else a contiguous copy."""
    if t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=CL)):
        return t
    if t.numel() and t.is_non_overlapping_and_dense():
        return t
    return t.contiguous()


class ActFn(torch.autograd.Function):
    """Stand-alone activation (Mish.forward, darknet/darknet.py:14-20): flat elementwise sweep."""

    @staticmethod
    def forward(ctx, x, act):
        L = lib()
        _require_gpu(x, f'{act} input')
        x = _dense(x)
        y = torch.empty_like(x, memory_format=torch.preserve_format)
        if x.numel():
            check(L.y4_act_fwd_f32(_ptr(x), _ptr(y), x.numel(), ACT_IDS[act], _stream()), 'act_fwd')
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, g):
        L = lib()
        (x,) = ctx.saved_tensors
        if g.stride() != x.stride():
            g = torch.empty_like(x, memory_format=torch.preserve_format).copy_(g)
        dx = torch.empty_like(x, memory_format=torch.preserve_format)
        if x.numel():
            check(L.y4_act_bwd_f32(_ptr(x), _ptr(g), _ptr(dx), x.numel(), ACT_IDS[ctx.act], _stream()), 'act_bwd')
        return dx, None


def bboxes_iou_raw(a, b, xyxy=True):
    """bboxes_iou, yolo/model/yololoss.py:16-91 (no gradient: the reference only uses it on detached boxes)."""
    L = lib()
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != 4 or b.shape[1] != 4:
        raise IndexError                                    # yololoss.py:39-40
    _require_gpu(a, 'bboxes_iou boxes_a')
    _require_gpu(b, 'bboxes_iou boxes_b')
    a = a.detach().contiguous()
    b = b.detach().contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), device=a.device, dtype=torch.float32)
    check(L.y4_bboxes_iou_f32(_ptr(a), a.shape[0], _ptr(b), b.shape[0], 1 if xyxy else 0, _ptr(out), _stream()),
          'bboxes_iou')
    return out


class Slot:
    """Non-tensor wrapper so that a destination slice can be handed to an autograd Function without becoming one of
    its differentiable inputs."""

    def __init__(self, t):
        self.t = t


# ------------------------------------------------------------------ YOLO head
class YoloDecodeTrainFn(torch.autograd.Function):
    """YOLOLayer.forward train branch, yolo/model/yololayer.py:88-145."""

    @staticmethod
    def forward(ctx, logits, anchors_wh, n_classes):
        L = lib()
        _require_gpu(logits, 'YOLOLayer input')
        B, ch, F, F2 = logits.shape
        if F != F2:
            raise Y4Error('YOLOLayer assumes square maps (yololayer.py:94)')
        n_ch = 5 + n_classes
        A = ch // n_ch
        logits, ldl = as_nhwc(logits, need_vec4=False)
        output = torch.empty((B, A, F, F, n_ch), device=logits.device, dtype=torch.float32)
        pred = torch.empty((B, A, F, F, 4), device=logits.device, dtype=torch.float32)
        anc = float_array([v for wh in anchors_wh for v in wh])
        check(L.y4_yolo_decode_train_f32(_ptr(logits), ldl, _ptr(output), _ptr(pred), B, F, A, n_classes, anc,
                                         _stream()), 'yolo_decode_train')
        ctx.save_for_backward(logits)
        ctx.meta = (B, F, A, n_classes, anchors_wh, ldl)
        return output, pred

    @staticmethod
    def backward(ctx, g_output, g_pred):
        L = lib()
        (logits,) = ctx.saved_tensors
        B, F, A, n_classes, anchors_wh, ldl = ctx.meta
        n_ch = 5 + n_classes
        if g_output is not None:
            g_output = g_output.contiguous()
        if g_pred is not None:
            g_pred = g_pred.contiguous()
        cpad = (A * n_ch + 31) // 32 * 32
        gl = empty_nhwc(B, A * n_ch, F, F, logits.device, pad_to=32)
        anc = float_array([v for wh in anchors_wh for v in wh])
        # g_logits pitch = cpad: pad channels are written as zeros by the kernel
        lg, ldl2 = as_nhwc(logits, need_vec4=False)
        if ldl2 != cpad:
            lg = empty_nhwc(B, A * n_ch, F, F, logits.device, pad_to=32)
            lg.copy_(logits)
            ldl2 = cpad
        check(L.y4_yolo_decode_bwd_f32(_ptr(lg), ldl2, _ptr(g_output), _ptr(g_pred), _ptr(gl), B, F, A, n_classes,
                                       anc, _stream()), 'yolo_decode_bwd')
        return gl, None, None


def yolo_decode_eval(logits, anchors_wh, n_classes, stride, out=None, n_total=None, box_off=0):
    """YOLOLayer.forward eval branch, yolo/model/yololayer.py:146-166; optionally writes straight
    into rows [box_off, box_off + A*F*F) of a shared [B, n_total, 5+C] buffer (the torch.cat of
    yolo/model/yolov4.py:324 without a copy)."""
    L = lib()
    _require_gpu(logits, 'YOLOLayer input')
    B, ch, F, _ = logits.shape
    n_ch = 5 + n_classes
    A = ch // n_ch
    logits, ldl = as_nhwc(logits, need_vec4=False)
    if out is None:
        n_total = A * F * F
        out = torch.empty((B, n_total, n_ch), device=logits.device, dtype=torch.float32)
        box_off = 0
    anc = float_array([v for wh in anchors_wh for v in wh])
    check(L.y4_yolo_decode_eval_f32(_ptr(logits), ldl, _ptr(out), n_total, box_off, B, F, A, n_classes, anc,
                                    float(stride), _stream()), 'yolo_decode_eval')
    return out


class YoloLossLayerFn(torch.autograd.Function):
    """YOLOLoss.build_target + loss terms for one layer, yolo/model/yololoss.py:118-371,385-432."""

    @staticmethod
    def forward(ctx, output, pred, labels, cfg):
        L = lib()
        _require_gpu(output, 'YOLOLoss output')
        B, A, F, _, n_ch = output.shape
        C = n_ch - 5
        K = labels.shape[1]
        if not output.is_contiguous():
            raise Y4Error('YOLOLoss: `output` must be contiguous [B,A,F,F,5+C]')
        pred = pred.detach().contiguous()
        labels = labels.detach().to(device=output.device, dtype=torch.float32).contiguous()
        nbytes = L.y4_yolo_loss_workspace(B, F, A, K, C)
        ws = _ws(nbytes, output.device)
        obj_mask = torch.empty((B, A, F, F), device=output.device, dtype=torch.float32)
        parts = torch.empty(4, device=output.device, dtype=torch.float64)
        anchors = float_array([v for wh in cfg['all_anchors'] for v in wh])
        amask = int_array(cfg['anch_mask'])
        check(L.y4_yolo_loss_fwd_f32(_ptr(output), _ptr(pred), _ptr(labels), K, B, F, A, C, float(cfg['stride']),
                                     float(cfg['ignore_thresh']), anchors, len(cfg['all_anchors']), amask,
                                     _ptr(obj_mask), _ptr(parts), _ptr(ws), nbytes, _stream()), 'yolo_loss_fwd')
        ctx.meta = (B, F, A, K, C, nbytes)
        ctx.ws = ws
        ctx.obj_mask = obj_mask
        ctx.output = output          # later mutated in place exactly like the reference does
        ctx.mutate = cfg.get('mutate_output', True)
        cfg['last'] = dict(obj_mask=obj_mask, parts=parts, ws=ws, nbytes=nbytes, dims=(B, F, A, K, C))
        loss = parts.sum().to(torch.float32)
        if ctx.mutate:
            # side effect of yololoss.py:402-407 on outputs[*]['output']; done through the raw pointer,
            # the backward kernel is told that the w/h channels already carry the scale factor
            check(L.y4_yolo_loss_mask_output_f32(_ptr(output), _ptr(obj_mask), B, F, A, K, C, _ptr(ws), nbytes,
                                                 _stream()), 'yolo_loss_mask_output')
        return loss

    @staticmethod
    def backward(ctx, gloss):
        L = lib()
        B, F, A, K, C, nbytes = ctx.meta
        g = torch.empty_like(ctx.output)
        gs = gloss.detach().to(torch.float32).reshape(1).contiguous()
        check(L.y4_yolo_loss_bwd_f32(_ptr(ctx.output), 1 if ctx.mutate else 0, _ptr(ctx.obj_mask), _ptr(gs), _ptr(g),
                                     B, F, A, K, C, _ptr(ctx.ws), nbytes, _stream()), 'yolo_loss_bwd')
        return g, None, None, None


def yolo_loss_dense_targets(last):
    """Materialise target / tgt_mask / tgt_scale (yololoss.py:173-187) from the sparse records."""
    L = lib()
    B, F, A, K, C = last['dims']
    dev = last['obj_mask'].device
    target = torch.empty((B, A, F, F, 5 + C), device=dev, dtype=torch.float32)
    tgt_mask = torch.empty((B, A, F, F, 4 + C), device=dev, dtype=torch.float32)
    tgt_scale = torch.empty((B, A, F, F, 2), device=dev, dtype=torch.float32)
    check(L.y4_yolo_loss_dense_targets_f32(_ptr(target), _ptr(tgt_mask), _ptr(tgt_scale), B, F, A, K, C,
                                           _ptr(last['ws']), last['nbytes'], _stream()), 'yolo_loss_dense_targets')
    return target, tgt_mask, tgt_scale
