# -*- coding: utf-8 -*-
"""Intermediate tensors under observation (VERDICT r3 missing #5 / ADVICE r3).  The reference's ConvBNAct returns a plain
fp32 tensor (darknet/darknet.py:53-58); here a training-mode layer whose only consumer is a plane-taking conv hands out a
pre-split tensor (two fp16 pieces per element).  The contract checked here: (1) a module with a forward hook -- and every
container the fast path would otherwise step over -- returns / receives real fp32 values, identical to the Y4_PLANES=0 run
to fp32 rounding, and its hooks fire exactly as often as in the reference's call sequence; (2) a pre-split tensor that does
escape refuses to be read as numbers."""
import pytest
import torch
from torch import nn

import recipe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    import yolov4_amd
    assert yolov4_amd.lib().y4_get_conv_mode() == 3
    return torch.device('cuda:0')


def _model(dev):
    from yolov4_amd.yolo.model.yolov4 import YOLOv4
    m = YOLOv4(recipe.MODEL_CFG, device=dev)
    sd = m.state_dict()
    recipe.fill_state_dict_(sd, 7)
    m.load_state_dict(sd)
    return m.to(dev).train()


def _loss(m, x, labels, dev):
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    crit = YOLOLoss(recipe.MODEL_CFG, 0.7, device=dev)
    for p in m.parameters():
        p.grad = None
    loss = crit(m(x), {'padded_labels': labels})
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach())


WATCHED = [
    'backbone.stage3.part2.1.module_list.0.0',      # 1x1 conv of a residual unit: sole consumer is the unit's 3x3 conv
    'backbone.stage3.part2.1.module_list.0',        # the nn.Sequential pair the fast path steps over
    'backbone.stage3.part2.1',                      # ResBlock: its result normally leaves pre-split for part2[2]
    'backbone.stage4.part2',                        # Sequential(ConvBNAct, ResBlock, ConvBNAct), unrolled by CSPDownSample.forward
    'backbone.stage4.base',                         # stride-2 conv read by the two 1x1 split convs
    'neck.spp',                                     # SPPBlock -> fpn.module1[0]
    'neck.fpn.module2',                             # five-conv chain
    'neck.pan',                                     # returns p3, whose only reader is head.yolo3[0]
    'head.yolo3',                                   # Sequential(conv, conv, YOLOLayer) stepped over by Head.forward
    'head.yolo3.0',
]


def test_forward_hooks_see_fp32_values_equal_to_the_planes_off_run(dev):
    from yolov4_amd import ops
    from yolov4_amd.ops import PlanesTensor
    m = _model(dev)
    x = recipe.randn((2, 3, 128, 128), 1).to(dev)
    labels = recipe.synth_labels(2, 128, 2, counts=[5, 9]).to(dev)
    mods = dict(m.named_modules())
    seen = {}

    def watch(name):
        def hook(mod, inp, out):
            for t in inp:
                assert not isinstance(t, PlanesTensor) and not getattr(t, 'y4_planes', False), name
            outs = out if isinstance(out, (tuple, list)) else [out]
            for t in outs:
                if torch.is_tensor(t):
                    assert not isinstance(t, PlanesTensor) and not getattr(t, 'y4_planes', False), name
            t = outs[-1]
            if isinstance(t, dict):
                t = t['output']
            seen.setdefault(name, []).append(t.detach().float().cpu().clone())
        return hook

    base = _loss(m, x, labels, dev)                          # unobserved: the fast path
    handles = [mods[n].register_forward_hook(watch(n)) for n in WATCHED]
    was = ops.PLANES['on']
    try:
        on = _loss(m, x, labels, dev)
        got_on = {k: v for k, v in seen.items()}
        g_on = [p.grad.clone() for p in m.parameters()]
        seen.clear()
        ops.PLANES['on'] = False
        off = _loss(m, x, labels, dev)
        got_off = {k: v for k, v in seen.items()}
        g_off = [p.grad.clone() for p in m.parameters()]
    finally:
        ops.PLANES['on'] = was
        for h in handles:
            h.remove()
    assert abs(on - off) <= 3e-4 * abs(off) and abs(base - off) <= 3e-4 * abs(off), (base, on, off)
    for n in WATCHED:
        assert len(got_on[n]) == 1 and len(got_off[n]) == 1, (n, len(got_on.get(n, [])))     # fired once, as in the reference
        a, b = got_on[n][0], got_off[n][0]
        assert a.shape == b.shape and bool(torch.isfinite(a).all())
        assert float((a - b).abs().max()) <= 2e-3 * max(float(b.abs().max()), 1e-6), n      # (B = 2 batch statistics: ill-conditioned)
    va, vb = torch.cat([g.flatten() for g in g_on]).double(), torch.cat([g.flatten() for g in g_off]).double()
    assert float((va * vb).sum() / (va.norm() * vb.norm())) > 0.999      # same gradient (conditioning: DESIGN section 2)
    # and with the hooks gone the fast path is back: the same layer hands out a pre-split tensor again
    blk = mods['backbone.stage3.part2.1'].module_list[0]
    from yolov4_amd.darknet.darknet import takes_planes
    z = blk[0](torch.randn(2, 128, 16, 16, device=dev).contiguous(memory_format=torch.channels_last), out_planes=takes_planes(blk[1]))
    assert isinstance(z, PlanesTensor)


def test_global_module_hook_keeps_every_intermediate_fp32(dev):
    from yolov4_amd.ops import PlanesTensor
    from yolov4_amd.darknet.darknet import ConvBNAct, chain
    torch.manual_seed(5)
    seq = nn.Sequential(ConvBNAct(128, 256, 3, 1), ConvBNAct(256, 128, 1, 1), ConvBNAct(128, 256, 3, 1)).to(dev).train()
    x = torch.randn(2, 128, 12, 12, device=dev).contiguous(memory_format=torch.channels_last)
    kinds = []
    h = torch.nn.modules.module.register_module_forward_hook(lambda mod, i, o: kinds.append(type(o)) if isinstance(mod, ConvBNAct) else None)
    try:
        y = chain(seq, x)
    finally:
        h.remove()
    assert len(kinds) == 3 and all(k is not PlanesTensor for k in kinds)
    kinds.clear()
    assert isinstance(seq[0](x, out_planes=True), PlanesTensor)          # unobserved again
    assert bool(torch.isfinite(y).all())


def test_pre_split_tensor_refuses_to_be_read_as_numbers(dev, tmp_path):
    from yolov4_amd._lib import Y4Error
    from yolov4_amd.darknet.darknet import ConvBNAct
    m = ConvBNAct(64, 128, 1, 1).to(dev).train()
    z = m(torch.randn(2, 64, 8, 8, device=dev).contiguous(memory_format=torch.channels_last), out_planes=True)
    assert type(z).__name__ == 'PlanesTensor' and z.requires_grad and z.grad_fn is not None
    for read in (lambda: z.cpu(), lambda: z.detach().cpu(), lambda: z.numpy(), lambda: z + 1, lambda: z[0], lambda: z.sum(),
                 lambda: float(z.max()), lambda: z.tolist(), lambda: z.to('cpu'), lambda: z.clone(),
                 lambda: torch.save(z, tmp_path / 'z.pt'), lambda: torch.save({'feat': z.detach()}, tmp_path / 'z2.pt')):
        with pytest.raises(Y4Error):
            read()
    assert 'NOT float32' in repr(z) and 'NOT float32' in f'{z}'
    # the consumer still takes it, forward and backward
    c = ConvBNAct(128, 128, 3, 1).to(dev).train()
    c(z).sum().backward()
    assert m.conv.weight.grad is not None and bool(torch.isfinite(m.conv.weight.grad).all())


@pytest.mark.parametrize('stage', ['stage1', 'stage3'])
def test_csp_fork_fan_in_folded_into_the_dgrad_epilogue(dev, stage):
    """The gradient fan-in of a CSP fork (darknet._FORK_FOLD): the split conv whose backward runs first parks its dx, the other
    adds it as the skip operand of its dgrad -- same sums as the separate add pass (one fp32 addition per element either way),
    for the streaming 1x1 kernels (stage 1) and the plane kernels (stage 3); and a hook on one of the two convs switches it off."""
    from yolov4_amd.darknet import darknet as D
    torch.manual_seed(9)
    blk = (D.CSPDownSample0(32, 64, 3, 2) if stage == 'stage1' else D.CSPDownSample(128, 256, 3, 2, num_blocks=2)).to(dev).train()
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.8, 1.2)
            nn.init.normal_(m.bias, 0, 0.1)
    cin, S = (32, 208) if stage == 'stage1' else (128, 40)
    x = torch.randn(2, cin, S, S, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(2, blk.transition.conv.out_channels, S // 2, S // 2, device=dev).contiguous(memory_format=torch.channels_last)

    def run(fold):
        D._FORK_FOLD = fold
        for p in blk.parameters():
            p.grad = None
        x.grad = None
        (blk(x) * w).sum().backward()
        torch.cuda.synchronize()
        return x.grad.clone(), [p.grad.clone() for p in blk.parameters()]
    was, was_cat = D._FORK_FOLD, D._CAT_PLANES
    try:
        D._CAT_PLANES = False                           # (a hook on part1 also switches the pre-split concat off: not this test's subject)
        gx1, gp1 = run(True)
        gx0, gp0 = run(False)
        h = blk.part1.register_forward_hook(lambda m, i, o: None)
        gxh, _ = run(True)                              # observed: no box, autograd's own fan-in
        h.remove()
    finally:
        D._FORK_FOLD, D._CAT_PLANES = was, was_cat
    assert float((gx1 - gx0).abs().max()) <= 2e-5 * float(gx0.abs().max())
    assert torch.equal(gxh, gx0)
    for a, b in zip(gp1, gp0):
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize('mode', ['f16x2', 'bf16'])
def test_csp_concat_written_pre_split_feeds_the_transition_conv_on_the_plane_kernels(dev, mode):
    """darknet._CAT_PLANES: the two producers of the concat in front of a CSP transition conv (darknet.py:154-163 in the reference)
    write their slots of the buffer pre-split under ONE joint scale (y4_bn_planes_bound_f32 chained), the concat-fed 1x1 conv
    runs forward / dgrad / wgrad on the DMA kernels -- against the same block with the switch off (fp32 concat, gather kernels):
    fp32-grade agreement in f16x2, bf16-grade in the bf16 mode; and a hook on the block switches it off."""
    import yolov4_amd
    from yolov4_amd import ops
    from yolov4_amd.darknet import darknet as D
    torch.manual_seed(11)
    blk = D.CSPDownSample(128, 256, 3, 2, num_blocks=2).to(dev).train()
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.8, 1.2)
            nn.init.normal_(m.bias, 0, 0.1)
    S = 40
    x = torch.randn(3, 128, S, S, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(3, 256, S // 2, S // 2, device=dev).contiguous(memory_format=torch.channels_last)
    seen = []

    def run(on):
        D._CAT_PLANES = on
        for p in blk.parameters():
            p.grad = None
        x.grad = None
        out = blk(x)
        seen.append(ops.last_conv_kernel())          # the block's last conv = the transition conv
        (out * w).sum().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in blk.parameters()]
    was, old_mode = D._CAT_PLANES, yolov4_amd.get_conv_mode()
    try:
        yolov4_amd.set_conv_mode(mode)
        o1, gx1, gp1 = run(True)
        o0, gx0, gp0 = run(False)
        h = blk.register_forward_hook(lambda m, i, o: None)
        oh, gxh, _ = run(True)                       # observed: the fp32 concat
        h.remove()
    finally:
        D._CAT_PLANES = was
        yolov4_amd.set_conv_mode(old_mode)
    assert 'conv_planes_mfma' in seen[0] and 'conv_planes_mfma' not in seen[1] and 'conv_planes_mfma' not in seen[2], seen
    tol_o, tol_g = (2e-5, 2e-4) if mode == 'f16x2' else (3e-2, 1.5e-1)
    assert float((o1 - o0).abs().max()) <= tol_o * float(o0.abs().max())
    assert float((gx1 - gx0).abs().max()) <= tol_g * float(gx0.abs().max())
    for a, b in zip(gp1, gp0):
        assert float((a - b).abs().max()) <= tol_g * max(float(b.abs().max()), 1e-6)
    assert torch.equal(oh, o0) and torch.equal(gxh, gx0)


@pytest.mark.parametrize('mode', ['f16x2', 'bf16'])
def test_fpn_concats_written_pre_split_through_the_upsample(dev, mode):
    """FPNBlock (yolov4.py:93-141 in the reference): cat([lateral(x), upsample(reduce(f))]) -- the lateral conv writes its slot
    of the concat buffer pre-split, the reducing conv writes planes under the buffer's joint scale (its bound over the SMALL
    map's pixel count) and the x2 upsample copies those pixel rows into the other slot; against the switch off."""
    import yolov4_amd
    from yolov4_amd import ops
    from yolov4_amd.darknet import darknet as D
    from yolov4_amd.yolo.model.yolov4 import FPNBlock
    torch.manual_seed(13)
    blk = FPNBlock().to(dev).train()
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.8, 1.2)
            nn.init.normal_(m.bias, 0, 0.1)
        if isinstance(m, D.ConvBNAct):
            # the block's LeakyReLU has a kink: a pre-activation within rounding of zero takes the other slope in the other arm,
            # and BatchNorm backward spreads that over its whole channel -- Mish here, so that the arms can be compared tightly
            m.act_name = 'mish'
    h = 9
    xs = [torch.randn(2, c, h * f, h * f, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
          for c, f in ((256, 4), (512, 2), (512, 1))]
    ws = [torch.randn(2, c, h * f, h * f, device=dev).contiguous(memory_format=torch.channels_last) for c, f in ((128, 4), (256, 2), (512, 1))]

    def run(on):
        D._CAT_PLANES = on
        for p in blk.parameters():
            p.grad = None
        for x in xs:
            x.grad = None
        outs = blk(*xs)
        sum((o * w).sum() for o, w in zip(outs, ws)).backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for o in outs], [x.grad.clone() for x in xs], [p.grad.clone() for p in blk.parameters()]
    was, old_mode = D._CAT_PLANES, yolov4_amd.get_conv_mode()
    try:
        yolov4_amd.set_conv_mode(mode)
        assert blk._cat_planes(blk.conv4, blk.conv3, blk.upsample1, blk.module2, xs[1], xs[2]) is not None
        o1, gx1, gp1 = run(True)
        o0, gx0, gp0 = run(False)
        hk = blk.upsample1.register_forward_hook(lambda m, i, o: None)
        assert blk._cat_planes(blk.conv4, blk.conv3, blk.upsample1, blk.module2, xs[1], xs[2]) is None
        hk.remove()
    finally:
        D._CAT_PLANES = was
        yolov4_amd.set_conv_mode(old_mode)
    if mode == 'f16x2':                              # fp32-grade either way
        for a, b in zip(o1, o0):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
        for a, b in zip(gx1 + gp1, gx0 + gp0):
            assert float((a - b).abs().max()) <= 3e-4 * max(float(b.abs().max()), 1e-6)
    else:
        # hybrid bf16 mode: with the switch on, the two concat-fed convs compute in bf16 instead of fp32-grade -- a bf16-grade
        # difference (mean error against the tensor's range; BatchNorm backward over a few hundred samples amplifies single
        # roundings, so no max-norm statement)
        for a, b in zip(o1 + gx1 + gp1, o0 + gx0 + gp0):
            assert float((a - b).abs().mean()) <= 2e-2 * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize('mode,k', [('f16x2', 3), ('f16x2', 1), ('bf16', 3)])
def test_head_output_conv_without_batchnorm_over_a_pre_split_input(dev, mode, k):
    """Head.yolo*[0:2] (yolov4.py:235-251 in the reference): ConvBNAct -> conv with bias, no BatchNorm, 255 output channels.
    With darknet._HEAD_PLANES the 3x3 conv's activation leaves pre-split and the output conv runs forward (bias in the
    skip-operand epilogue), dgrad and wgrad on the plane kernels, its 255 output channels padded with a zero filter to 256
    and its gradient split by one extra pass -- against the switch off (fp32 activation, halo / gather kernels)."""
    import yolov4_amd
    from yolov4_amd import ops
    from yolov4_amd.darknet import darknet as D
    torch.manual_seed(17)
    a = D.ConvBNAct(128, 256, 3, 1, act='leaky_relu').to(dev).train()
    b = D.ConvBNAct(256, 255, k, 1, bias=True, bn=False, act='linear').to(dev).train()
    nn.init.uniform_(a.norm.weight, 0.8, 1.2)
    nn.init.normal_(a.norm.bias, 0, 0.1)
    nn.init.normal_(b.conv.bias, 0, 0.5)
    x = torch.randn(3, 128, 21, 21, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(3, 255, 21, 21, device=dev).contiguous(memory_format=torch.channels_last)
    kinds = []

    def run(on):
        D._HEAD_PLANES = on
        for p in list(a.parameters()) + list(b.parameters()):
            p.grad = None
        x.grad = None
        want = D.takes_planes(b, geo=(3, 21, 21))
        z = a(x, out_planes=bool(want))
        kinds.append(type(z).__name__)
        out = b(z)
        (out * w).sum().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in list(a.parameters()) + list(b.parameters())]
    was, old_mode = D._HEAD_PLANES, yolov4_amd.get_conv_mode()
    try:
        yolov4_amd.set_conv_mode(mode)
        o1, gx1, gp1 = run(True)
        o0, gx0, gp0 = run(False)
    finally:
        D._HEAD_PLANES = was
        yolov4_amd.set_conv_mode(old_mode)
    assert kinds == ['PlanesTensor', 'Tensor']
    assert o1.shape == o0.shape == (3, 255, 21, 21)
    if mode == 'f16x2':
        assert float((o1 - o0).abs().max()) <= 2e-5 * float(o0.abs().max())
        for p1, p0 in zip([gx1] + gp1, [gx0] + gp0):
            assert p1.shape == p0.shape
            assert float((p1 - p0).abs().max()) <= 2e-4 * max(float(p0.abs().max()), 1e-6)
    else:
        for p1, p0 in zip([o1, gx1] + gp1, [o0, gx0] + gp0):
            assert p1.shape == p0.shape
            assert float((p1 - p0).abs().mean()) <= 2e-2 * max(float(p0.abs().max()), 1e-6)


@pytest.mark.parametrize('mode', ['f16x2', 'bf16'])
def test_pan_concats_take_the_lateral_tensor_by_a_split_instead_of_a_copy(dev, mode):
    """PANBlock (yolov4.py:160-189 in the reference): cat([down(p), f]) -- the stride-2 conv writes its slot of the concat buffer
    pre-split, the lateral fp32 tensor f is SPLIT into the other slot (y4_planes_split_into_f32) under the joint scale
    max(bound of the conv's BatchNorm, max|f|); the concat-fed 1x1 conv then runs on the DMA kernels.  Against the switch off."""
    import yolov4_amd
    from yolov4_amd import ops
    from yolov4_amd.darknet import darknet as D
    from yolov4_amd.yolo.model.yolov4 import PANBlock
    torch.manual_seed(15)
    blk = PANBlock().to(dev).train()
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.8, 1.2)
            nn.init.normal_(m.bias, 0, 0.1)
        if isinstance(m, D.ConvBNAct):
            m.act_name = 'mish'                      # (smooth activation: see the FPN test)
    h = 10
    xs = [(3.0 * torch.randn(2, c, h * f, h * f, device=dev)).contiguous(memory_format=torch.channels_last).requires_grad_(True)
          for c, f in ((128, 4), (256, 2), (512, 1))]           # (laterals with a range well above the BatchNorm outputs')
    ws = [torch.randn(2, c, h * f, h * f, device=dev).contiguous(memory_format=torch.channels_last) for c, f in ((128, 4), (256, 2), (512, 1))]

    def run(on):
        D._CAT_PLANES = on
        for p in blk.parameters():
            p.grad = None
        for x in xs:
            x.grad = None
        outs = blk(*xs)
        sum((o * w).sum() for o, w in zip(outs, ws)).backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for o in outs], [x.grad.clone() for x in xs], [p.grad.clone() for p in blk.parameters()]
    was, old_mode = D._CAT_PLANES, yolov4_amd.get_conv_mode()
    try:
        yolov4_amd.set_conv_mode(mode)
        assert blk._cat_planes(blk.conv1, blk.module1, xs[1]) is not None
        o1, gx1, gp1 = run(True)
        o0, gx0, gp0 = run(False)
    finally:
        D._CAT_PLANES = was
        yolov4_amd.set_conv_mode(old_mode)
    if mode == 'f16x2':
        for a, b in zip(o1, o0):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
        for a, b in zip(gx1 + gp1, gx0 + gp0):
            assert float((a - b).abs().max()) <= 3e-4 * max(float(b.abs().max()), 1e-6)
    else:
        for a, b in zip(o1 + gx1 + gp1, o0 + gx0 + gp0):
            assert float((a - b).abs().mean()) <= 2e-2 * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize('mode', ['f16x2', 'bf16'])
def test_spp_concat_leaves_pre_split_for_its_conv(dev, mode):
    """SPPBlock (yolov4.py:52-77 in the reference): cat([pool5, pool9, pool5, x]) has one reader, the 1x1 conv 2048 -> 512: the
    concatenated tensor is split once (scale: the maximum of x, which pooling cannot exceed) and the conv runs on the DMA
    kernels.  Against the switch off."""
    import yolov4_amd
    from yolov4_amd import ops
    from yolov4_amd.darknet import darknet as D
    from yolov4_amd.yolo.model.yolov4 import SPPBlock
    torch.manual_seed(19)
    blk = SPPBlock().to(dev).train()
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.8, 1.2)
            nn.init.normal_(m.bias, 0, 0.1)
        if isinstance(m, D.ConvBNAct):
            m.act_name = 'mish'                      # (smooth activation: see the FPN test)
    x = torch.randn(3, 1024, 13, 13, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(3, 512, 13, 13, device=dev).contiguous(memory_format=torch.channels_last)
    seen = []

    def run(on):
        D._CAT_PLANES = on
        for p in blk.parameters():
            p.grad = None
        x.grad = None
        out = blk(x)
        seen.append(ops.last_conv_kernel())
        (out * w).sum().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in blk.parameters()]
    was, old_mode = D._CAT_PLANES, yolov4_amd.get_conv_mode()
    try:
        yolov4_amd.set_conv_mode(mode)
        o1, gx1, gp1 = run(True)
        o0, gx0, gp0 = run(False)
    finally:
        D._CAT_PLANES = was
        yolov4_amd.set_conv_mode(old_mode)
    assert 'conv_planes_mfma' in seen[0] and 'conv_planes_mfma' not in seen[1], seen
    if mode == 'f16x2':
        assert float((o1 - o0).abs().max()) <= 2e-5 * float(o0.abs().max())
        for a, b in zip([gx1] + gp1, [gx0] + gp0):
            assert float((a - b).abs().max()) <= 3e-4 * max(float(b.abs().max()), 1e-6)
    else:
        for a, b in zip([o1, gx1] + gp1, [o0, gx0] + gp0):
            assert float((a - b).abs().mean()) <= 2e-2 * max(float(b.abs().max()), 1e-6)


def test_bf16_mode_runs_the_64_channel_1x1_layers_of_stage_1_on_the_plane_kernels(dev):
    """darknet._BF16_N64 (conv mode 'bf16' only): CSPDownSample0's split convs, its transition and the conv behind its residual unit
    are 1x1 layers with 64 output channels -- half a column tile -- and run on the bf16 plane kernels all the same, so that their
    activations, gradients and conv results pass the BatchNorm sweeps as bf16.  Against the switch off (fp32-grade kernels):
    a bf16-grade difference; in the fp32-grade mode the switch changes nothing."""
    import yolov4_amd
    from yolov4_amd import ops
    from yolov4_amd.darknet import darknet as D
    torch.manual_seed(23)
    blk = D.CSPDownSample0(32, 64, 3, 2).to(dev).train()
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.8, 1.2)
            nn.init.normal_(m.bias, 0, 0.1)
    x = torch.randn(2, 32, 48, 48, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(2, 64, 24, 24, device=dev).contiguous(memory_format=torch.channels_last)
    seen = []

    def run(on):
        D._BF16_N64 = on
        for p in blk.parameters():
            p.grad = None
        x.grad = None
        out = blk(x)
        seen.append(ops.last_conv_kernel())          # the transition conv
        (out * w).sum().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in blk.parameters()]
    was, old_mode = D._BF16_N64, yolov4_amd.get_conv_mode()
    try:
        yolov4_amd.set_conv_mode('bf16')
        assert D.takes_planes(blk.part1) and D.takes_planes(blk.transition) and not D.takes_planes(blk.part2_1_2[0])
        o1, gx1, gp1 = run(True)
        o0, gx0, gp0 = run(False)
        yolov4_amd.set_conv_mode('f16x2')
        assert not D.takes_planes(blk.part1)
        f1 = run(True)
        f0 = run(False)
    finally:
        D._BF16_N64 = was
        yolov4_amd.set_conv_mode(old_mode)
    assert 'conv_planes_mfma' in seen[0] and 'conv_planes_mfma' not in seen[1], seen
    for a, b in zip([o1, gx1] + gp1, [o0, gx0] + gp0):
        assert float((a - b).abs().mean()) <= 2e-2 * max(float(b.abs().max()), 1e-6)
    assert torch.equal(f1[0], f0[0]) and torch.equal(f1[1], f0[1])
