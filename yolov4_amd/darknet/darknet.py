# -*- coding: utf-8 -*-
"""CSPDarknet53 building blocks behind the reference's module API
(darknet/darknet.py:14-138 of zjykzj/YOLOv4), executed by libyolov4_amd.so.

`conv` / `norm` / `act` sub-modules exist only as parameter containers with the
reference's attribute names (so state_dict keys and shapes are identical); their
own forward() is never called -- ConvBNAct.forward issues one fused library
sequence instead: implicit-GEMM conv (MFMA) -> batch statistics -> normalise +
activation (+ ResBlock skip), or in eval mode a single conv kernel with the
folded BatchNorm, activation and skip in its epilogue.
"""
import os

import torch
from torch import nn

from .. import ops


class Mish(nn.Module):
    """x * tanh(softplus(x)), darknet/darknet.py:14-20.  Inside ConvBNAct the arithmetic is fused into the BN / conv
    epilogue kernels (this module is then only a marker); called on its own it runs the same device function as a
    flat elementwise kernel (y4_act_fwd_f32 / y4_act_bwd_f32)."""

    def forward(self, x):
        return ops.ActFn.apply(x, 'mish')


_ACT_MODULES = {
    'relu': lambda: nn.ReLU(inplace=True),
    'leaky_relu': lambda: nn.LeakyReLU(negative_slope=0.1, inplace=True),
    'mish': Mish,
    'linear': nn.Identity,
}


class ConvBNAct(nn.Module):

    def __init__(self, in_ch: int, out_ch: int, kernel_size: int, stride: int, bias=False, bn=True, act='leaky_relu'):
        super().__init__()
        if act not in _ACT_MODULES:
            raise ValueError(f"{act} does not support.")
        pad = (kernel_size - 1) // 2
        self.conv = nn.Conv2d(in_ch, out_ch, (kernel_size, kernel_size), (stride, stride), padding=pad, bias=bias)
        # KRSC in memory, OIHW in the state_dict
        self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)
        self.norm = nn.BatchNorm2d(out_ch) if bn else nn.Identity()
        self.act = _ACT_MODULES[act]()
        self.act_name = act
        self.kernel_size = kernel_size
        self.stride = stride
        self.has_bn = bool(bn)

    def forward(self, x, residual=None, out=None, dres_put=None, dres_take=None, out_planes=False, dx_put=None, scale_from=None):
        """out: optional destination (a CatBuffer slot) for the activation; dres_put / dres_take: the shared box through
        which a ResBlock unit's 3x3 conv hands the skip gradient to its 1x1 conv (see ResBlock); out_planes: the caller
        guarantees that the SOLE consumer of the result is a ConvBNAct for which `takes_planes()` holds, so the
        activation may leave pre-split for the DMA-fed conv kernels (csrc/conv_planes.hip) instead of as fp32;
        out_planes='both': such a consumer exists beside fp32 ones -- the result is fp32 and carries a pre-split twin
        (tensor attribute y4_twin, handed on by ops.fork); scale_from: a pre-split ops.CatBuffer whose joint scale a planes-only
        result is written under (it will be copied into that buffer by an Upsample).  The reference has none of these arguments."""
        n = self.norm
        io = {}
        out_planes = soft(out_planes, self)
        cfg = {'out': out, 'k': self.kernel_size, 's': self.stride, 'act': self.act_name, 'bn': self.has_bn,
               'training': self.training, 'io': io, 'x_amax': ops.amax_of(x), 'out_amax': ops.amax_of(out),
               'dres_put': dres_put, 'dres_take': dres_take, 'out_planes': out_planes, 'dx_put': dx_put,
               'out_cat': getattr(out, 'y4_cat', None), 'scale_from': scale_from,
               'x_twin': getattr(x, 'y4_twin', None) if takes_planes(self, x.shape[2:], geo_of(x)) else None,
               'grad': torch.is_grad_enabled()}      # (autograd.Function.forward itself always runs with grad mode off)
        if self.has_bn:
            use_batch_stats = self.training or n.running_mean is None
            cfg['training'] = use_batch_stats
            cfg['eps'] = n.eps
            # nn.BatchNorm2d: momentum None = cumulative average
            cfg['momentum'] = n.momentum if n.momentum is not None else 1.0 / float(int(n.num_batches_tracked) + 1)
            track = self.training and n.track_running_stats
            cfg['running_mean'] = n.running_mean if (track or not use_batch_stats) else None
            cfg['running_var'] = n.running_var if (track or not use_batch_stats) else None
            cfg['nbt'] = n.num_batches_tracked if track else None
            gamma, beta = n.weight, n.bias
            cfg['gamma_param'], cfg['beta_param'] = n.weight, n.bias
        else:
            gamma = beta = None
        w = self.conv.weight
        if w.dim() == 4 and not w.is_contiguous(memory_format=torch.channels_last):
            # e.g. after load_state_dict into a freshly built module on another device
            self.conv.weight.data = w.data.contiguous(memory_format=torch.channels_last)
            w = self.conv.weight
        cfg['weight_param'] = self.conv.weight       # its gradient may be produced on the side stream (ops._wgrad_to_param)
        z = ops.ConvBNActFn.apply(x, w, self.conv.bias, gamma, beta, residual, cfg)
        if io.get('z_planes'):
            z = ops.as_planes(z)                     # (tags do not survive autograd's output wrapping: set on the result)
        if io.get('z_twin') is not None:
            z.y4_twin = io['z_twin']
        return ops.tag_amax(z, io.get('z_amax'))


def observed(*mods):
    """True if somebody may look at what one of these modules receives or returns: a forward / backward (pre-)hook on it,
    or a global module hook (torch.nn.modules.module.register_module_forward_hook ...).  Observed modules never see or
    return a pre-split tensor: producers write fp32 with a pre-split twin ('both') instead, hooked containers that the fast
    path would step over (nn.Sequential pairs and chains) are CALLED, so that their hooks fire as in the reference."""
    g = torch.nn.modules.module
    if (g._global_forward_hooks or g._global_forward_pre_hooks or g._global_backward_hooks or g._global_backward_pre_hooks
            or getattr(g, '_global_forward_hooks_always_called', None)):
        return True
    for m in mods:
        if m is not None and (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks):
            return True
    return False


def soft(want, *mods):
    """out_planes request `want`, downgraded from planes-only (True) to fp32 + twin ('both') when one of `mods` -- the
    modules through whose interface the tensor will pass -- is observed."""
    return 'both' if (want is True and observed(*mods)) else want


def takes_planes(m, hw=None, geo=None):
    """True if ConvBNAct `m` can consume a pre-split (planes) input right now: training-mode BatchNorm, conv mode 3 (or 2),
    whole 32-channel K tiles on both sides and at least one full 128-column tile of output channels; stride 1 (all three
    of its convs then run on the DMA kernels of csrc/conv_planes.hip), or a 3x3 stride-2 layer on an even map `hw` = (H, W)
    of its input (forward and wgrad on the DMA kernels, dgrad on the register-staged parity-class kernel).
    geo = (B, H, W) of its input, where the caller knows it: the operands must also fit the kernels' 32-bit buffer windows
    (ops.planes_fit) -- a producer must never emit planes its consumer cannot address, there is no fp32 form to fall back to."""
    if isinstance(m, ConvBNAct) and not m.has_bn:
        return _nobn_takes_planes(m, geo)
    if not isinstance(m, ConvBNAct) or not m.has_bn or not m.training:
        return False
    if geo is not None and hw is None and m.stride == 1:
        hw = geo[1:]
    if m.stride != 1 and not (hw is not None and ops.planes_stride2_ok(m.kernel_size, m.stride, int(hw[0]), int(hw[1]))):
        return False
    ci, co = m.conv.in_channels, m.conv.out_channels
    pm = ops.planes_mode()
    # at least one full 128-column tile of output channels -- except the 1x1 layers with 64 output channels in the bf16 mode
    # (stage 1 / 2: split convs, transition, the 1x1 of the stage-2 residual units): one MFMA per product makes the half-empty
    # tile cheap, and what counts there is bytes -- bf16 activations, gradients and conv results through the BatchNorm sweeps
    # (699 -> 716 img/s with the 128 -> 64 layers, -> 727 with the 64 -> 64 ones; the 3x3 64 -> 64 layers: +-0, left alone)
    co_min = 64 if (pm == 'bf16' and _BF16_N64 and m.kernel_size == 1 and m.stride == 1) else 128
    if ci % 32 or co % 32 or co < co_min or ci < 64 or m.kernel_size not in (1, 3):
        return False
    if not (ops.PLANES['on'] and m.conv.weight.is_cuda):
        return False
    if not (pm == 'f16x2' or (pm == 'bf16' and ci % 64 == 0 and co % 64 == 0)):  # bf16 rows hold 64 channels
        return False
    # stride 2: backward picks the register-staged dgrad by itself when the plane one does not fit (ConvBNActFn.backward)
    return geo is None or ops.planes_fit(geo[0], geo[1], geo[2], ci, co, m.kernel_size, m.stride, dgrad=m.stride == 1)


# The head's output convs (no BatchNorm, bias, linear; yolo/model/yolov4.py:235-251 in the reference) over a pre-split input
# (Y4_HEAD_PLANES=0: off)
_HEAD_PLANES = os.environ.get('Y4_HEAD_PLANES', '1') != '0'
# bf16 mode: the 1x1 layers with 64 output channels on the plane kernels too (Y4_BF16_N64=0: off)
_BF16_N64 = os.environ.get('Y4_BF16_N64', '1') != '0'


def _nobn_takes_planes(m, geo):
    """A training-mode conv WITHOUT BatchNorm (bias, linear, stride 1) can take a pre-split input when autograd is on: its
    output channels are padded to whole K tiles inside ConvBNActFn (255 -> 256), its gradient is split by one extra pass."""
    if not (_HEAD_PLANES and m.training and torch.is_grad_enabled() and m.act_name == 'linear' and m.stride == 1
            and m.kernel_size in (1, 3) and ops.PLANES['on'] and m.conv.weight.is_cuda):
        return False
    pm = ops.planes_mode()
    if pm is None:
        return False
    q = 64 if pm == 'bf16' else 32
    ci, co = m.conv.in_channels, m.conv.out_channels
    cop = (co + q - 1) // q * q
    if ci % q or ci < 64 or cop < 128:
        return False
    return geo is None or ops.planes_fit(geo[0], geo[1], geo[2], ci, cop, m.kernel_size, 1, dgrad=True)


def geo_of(x, after=None):
    """(B, H, W) of NCHW tensor x, or of the result of ConvBNAct `after` applied to it."""
    st = after.stride if isinstance(after, ConvBNAct) else 1
    return (int(x.shape[0]), -(-int(x.shape[2]) // st), -(-int(x.shape[3]) // st))      # (k = 3, pad 1 or k = 1: ceil)


def plan_for(consumers, hw, batch=None):
    """out_planes request for a tensor of spatial size hw (batch size `batch`) read by the ConvBNAct modules `consumers`: True if
    its single reader takes planes, 'both' if one of several does (or somebody is looking), else False."""
    geo = (int(batch), int(hw[0]), int(hw[1])) if batch is not None else None
    takers = [m for m in consumers if takes_planes(m, hw, geo)]
    if not takers:
        return False
    if len(consumers) == 1:
        return soft(True, consumers[0])
    return 'both'


def chain(seq, x, last=False):
    """nn.Sequential of ConvBNAct layers, each feeding only the next: intermediates leave pre-split where the consumer
    can take them (same results as seq(x); the reference calls the Sequential).  last: out_planes of the final layer
    (True: its sole consumer takes planes; 'both': one of several does)."""
    if observed(seq):
        return seq(x)                                # hooks on the Sequential itself: the plain call (all-fp32 intermediates)
    mods = list(seq)
    for i, m in enumerate(mods):
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        want = soft(takes_planes(nxt, geo=geo_of(x, m)), nxt) if nxt is not None else last
        x = m(x, out_planes=want) if isinstance(m, ConvBNAct) else m(x)
    return x


def res_unit(pair, x, out_planes=False):
    """x + conv3x3(conv1x1(x)) (darknet.py:76-80): the skip is added in the 3x3's BN+act kernel; in backward the gradient
    arriving over the skip is parked by the 3x3 and added in the 1x1's dgrad epilogue, so neither direction spends a
    separate elementwise pass.  out_planes: as ConvBNAct.forward, for the unit's result."""
    xa, xb = ops.fork(x)
    if observed(pair):
        return ops.AddFn.apply(xb, pair(xa))         # hooks on the pair: the reference's x + module(x), unfused
    box = {} if (torch.is_grad_enabled() and xa.requires_grad and pair[0].training) else None
    return pair[1](pair[0](xa, dres_take=box, out_planes=soft(takes_planes(pair[1], geo=geo_of(xa)), pair[1])), residual=xb, dres_put=box,
                   out_planes=out_planes)


# A residual unit's result feeds the next unit's 1x1 conv AND its skip: fp32 for the skip plus a pre-split twin for the conv
# ('both') puts those 1x1 convs on the plane kernels at the price of one more 4-B/element write per unit (Y4_TWIN_RES=0: off)
_TWIN_RES = os.environ.get('Y4_TWIN_RES', '1') != '0'
# The gradient fan-in of a CSP fork (the stride-2 conv's result feeds the two 1x1 split convs): the split conv whose backward
# runs first parks its dx, the other adds it in its dgrad epilogue instead of a separate add pass (Y4_FORK_FOLD=0: off)
_FORK_FOLD = os.environ.get('Y4_FORK_FOLD', '1') != '0'
# The concat-fed CSP transition convs on the plane kernels (their producers write the concat buffer pre-split; Y4_CAT_PLANES=0: off)
_CAT_PLANES = os.environ.get('Y4_CAT_PLANES', '1') != '0'


def fork_box(x, *convs):
    """Shared box for two ConvBNAct readers of x (see _FORK_FOLD), or None when the fold does not apply."""
    if not (_FORK_FOLD and torch.is_grad_enabled() and x.requires_grad and not observed(*convs)):
        return None
    if not all(isinstance(m, ConvBNAct) and m.training and m.has_bn and m.kernel_size == 1 and m.stride == 1 for m in convs):
        return None
    return {}


class ResBlock(nn.Module):

    def __init__(self, ch, num_blocks=1, shortcut=True, act="mish"):
        super().__init__()
        self.shortcut = shortcut
        self.module_list = nn.ModuleList(
            nn.Sequential(ConvBNAct(ch, ch, 1, 1, act=act), ConvBNAct(ch, ch, 3, 1, act=act))
            for _ in range(num_blocks))

    def forward(self, x, out_planes=False):
        """out_planes: for the block's result (the reference's forward has no such argument)."""
        n = len(self.module_list)
        for i, pair in enumerate(self.module_list):
            if self.shortcut:
                if i + 1 < n:
                    want = 'both' if (_TWIN_RES and takes_planes(self.module_list[i + 1][0], geo=geo_of(x))) else False
                else:
                    want = soft(out_planes, self)
                x = res_unit(pair, x, out_planes=want)
            else:
                x = pair[1](pair[0](x))
        return x

    def first_takes_planes(self, geo=None):
        return _TWIN_RES and takes_planes(self.module_list[0][0], geo=geo)


class CSPDownSample0(nn.Module):

    def __init__(self, in_ch=32, out_ch=64, kernel_size=3, stride=2, act='mish'):
        super().__init__()
        self.base = ConvBNAct(in_ch, out_ch, kernel_size, stride, act=act)
        self.part1 = ConvBNAct(out_ch, out_ch, 1, 1, act=act)
        self.part2_1_1 = ConvBNAct(out_ch, out_ch, 1, 1, act=act)
        self.part2_1_2 = nn.Sequential(ConvBNAct(out_ch, out_ch // 2, 1, 1, act=act),
                                       ConvBNAct(out_ch // 2, out_ch, 3, 1, act=act))
        self.part2_2 = ConvBNAct(out_ch, out_ch, 1, 1, act=act)
        self.transition = ConvBNAct(out_ch * 2, out_ch, 1, 1, act=act)

    def forward(self, x, readers=None):
        """readers: the ConvBNAct modules that will read the result (the reference's forward has no such argument): where they
        take pre-split inputs the transition conv writes its result that way."""
        go = geo_of(x, self.base)
        both = soft(bool(_TWIN_RES and takes_planes(self.part1, geo=go) and takes_planes(self.part2_1_1, geo=go)),
                    self.part1, self.part2_1_1)
        xa, xb = ops.fork(self.base(x, out_planes=both))
        # (the concat in front of the transition conv pre-split where that conv takes planes: as CSPDownSample.forward)
        cat_planes = (_CAT_PLANES and takes_planes(self.transition, geo=geo_of(xa)) and self.part1.training and self.part2_2.training
                      and self.part1.has_bn and self.part2_2.has_bn
                      and not observed(self, self.part1, self.part2_2, self.transition))
        cb = ops.cat_buffer(xa, [self.part2_2.conv.out_channels, self.part1.conv.out_channels],
                            planes_norms=(self.part2_2.norm, self.part1.norm) if cat_planes else None)
        fb = fork_box(xa, self.part1, self.part2_1_1)
        x1 = self.part1(xa, out=cb.slot(1), dres_take=fb)
        x2 = res_unit(self.part2_1_2, self.part2_1_1(xb, dx_put=fb), out_planes=soft(takes_planes(self.part2_2, geo=go), self.part2_2))
        x2 = self.part2_2(x2, out=cb.slot(0))
        return self.transition(ops.cat([x2, x1], into=cb), out_planes=soft(plan_for(readers, xa.shape[2:], xa.shape[0]), self) if readers else False)


class CSPDownSample(nn.Module):

    def __init__(self, in_ch=64, out_ch=128, kernel_size=3, stride=2, num_blocks=1, shortcut=True, act='mish'):
        super().__init__()
        self.base = ConvBNAct(in_ch, out_ch, kernel_size, stride, act=act)
        self.part1 = ConvBNAct(out_ch, out_ch // 2, 1, 1, act=act)
        self.part2 = nn.Sequential(ConvBNAct(out_ch, out_ch // 2, 1, 1, act=act),
                                   ResBlock(out_ch // 2, num_blocks=num_blocks, shortcut=shortcut, act=act),
                                   ConvBNAct(out_ch // 2, out_ch // 2, 1, 1, act=act))
        self.transition = ConvBNAct(out_ch, out_ch, 1, 1, act=act)

    def forward(self, x, readers=None):
        """readers: as CSPDownSample0.forward."""
        # both consumers of the stride-2 conv's result are 1x1 convs: where both take planes the result leaves pre-split only
        go = geo_of(x, self.base)
        both = soft(bool(_TWIN_RES and takes_planes(self.part1, geo=go) and takes_planes(self.part2[0], geo=go)),
                    self.part1, self.part2, self.part2[0])
        xa, xb = ops.fork(self.base(x, out_planes=both))
        # the concat in front of the transition conv: where that conv takes planes and nobody is looking, its two producers'
        # BatchNorm sweeps write their slots of the buffer pre-split (one joint scale, ops.CatBuffer) and the concat-fed 1x1
        # conv runs forward, dgrad and wgrad on the DMA kernels instead of the register-staged gather kernels
        cat_planes = (_CAT_PLANES and takes_planes(self.transition, geo=go) and self.part1.has_bn and self.part2[2].has_bn
                      and self.part1.training and self.part2[2].training
                      and not observed(self, self.part1, self.part2, self.part2[2], self.transition))
        cb = ops.cat_buffer(xa, [self.part2[2].conv.out_channels, self.part1.conv.out_channels],
                            planes_norms=(self.part2[2].norm, self.part1.norm) if cat_planes else None)
        fb = None if observed(self.part2) else fork_box(xa, self.part1, self.part2[0])
        x1 = self.part1(xa, out=cb.slot(1), dres_take=fb)
        if observed(self.part2):
            x2 = self.part2(xb)                      # hooks on the Sequential: the plain call; cat copies its result in
            return self.transition(ops.cat([x2, x1], into=cb), out_planes=soft(plan_for(readers, xa.shape[2:], xa.shape[0]), self) if readers else False)
        blk = self.part2[1]
        # part2[0]'s result feeds the first unit's 1x1 conv and its skip; the block's result feeds part2[2] alone
        first = 'both' if (blk.shortcut and blk.first_takes_planes(go)) else False
        last = soft(bool(blk.shortcut and _TWIN_RES and takes_planes(self.part2[2], geo=go)), blk, self.part2[2])
        x2 = blk(self.part2[0](xb, out_planes=first, dx_put=fb), out_planes=last)
        x2 = self.part2[2](x2, out=cb.slot(0))
        return self.transition(ops.cat([x2, x1], into=cb), out_planes=soft(plan_for(readers, xa.shape[2:], xa.shape[0]), self) if readers else False)
