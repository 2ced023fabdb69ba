# -*- coding: utf-8 -*-
"""GPU parity, second batch: memory flatness of the training step, the stand-alone forms of fused pieces
(bboxes_iou, Mish, general nearest Upsample) against the reference's fixtures / torch CPU, BASELINE configs[0]
(1x3x416x416 eval forward vs the oracle), whole-model bf16 mode with a stated tolerance (configs[4]), the stored
backbone / neck gradients of the train-step fixture under a conditioning-aware bound, multi-tensor optimizers,
the drop-in BucketedDDP loop, and the packaged train / validate harness."""
import copy
import gc
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import recipe
from oracle import head as H
from oracle import network as NW

pytestmark = pytest.mark.gpu

CFG = recipe.MODEL_CFG
OPT_CFG = dict(recipe.FULL_CFG)
OPT_CFG['OPTIMIZER'] = {'TYPE': 'ADAM', 'LR': '3e-4', 'NO_BIAS': True, 'NO_NORM': True, 'MOMENTUM': 0.9, 'DECAY': 1e-5}
OPT_CFG['LR_SCHEDULER'] = {'TYPE': 'MultiStepLR', 'MILESTONES': [60, 90, 110], 'GAMMA': 0.1, 'IS_WARMUP': True,
                           'WARMUP_EPOCH': 5, 'MULTIPLIER': 1.0}
OPT_CFG['TRAIN'] = {'IMGSIZE': 608, 'MAX_EPOCHS': 120, 'ACCUMULATION_STEPS': 2}


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    import yolov4_amd
    assert yolov4_amd.lib().y4_device_count() >= 1
    return torch.device('cuda:0')


def close(a, b, atol=1e-4, rtol=1e-4, scale=True):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    a = a.astype(np.float64); b = b.astype(np.float64)
    s = max(1.0, float(np.abs(b).max())) if scale else 1.0
    np.testing.assert_allclose(a, b, atol=atol * s, rtol=rtol)


def _model(dev, seed=1234):
    from yolov4_amd.yolo.model.yolov4 import YOLOv4
    m = YOLOv4(CFG, device=dev)
    sd = m.state_dict()
    recipe.fill_state_dict_(sd, seed)
    m.load_state_dict(sd)
    return m.to(dev)


def _unwrap(ddp):
    for h in ddp._hooks:
        h.remove()
    for p in ddp.module.parameters():
        p.grad = None
        for a in ('_y4_grad_fresh', '_y4_grad_ready', '_y4_ddp'):
            if hasattr(p, a):
                delattr(p, a)


# ------------------------------------------------------------------ memory
def test_training_step_memory_is_flat(dev):
    """output -> grad_fn -> ctx -> cfg['out'] -> output kept every concat buffer alive until a
    full gc.collect() (+4.96 GiB per step at bs = 64).  With the collector OFF, allocated memory after step 3 must
    equal allocated memory after step 8: plain autograd frees per step (yolo/engine/build.py:59-69)."""
    from yolov4_amd.ddp import BucketedDDP
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    m = _model(dev).train()
    ddp = BucketedDDP(m)
    crit = YOLOLoss(CFG, 0.7, device=dev)
    x = recipe.randn((4, 3, 256, 256), 5).to(dev)
    labels = recipe.synth_labels(4, 256, 6).to(dev)
    gc.collect()
    gc.disable()
    try:
        marks = []
        for step in range(8):
            ddp.zero_grad()
            loss = crit(ddp(x), {'padded_labels': labels})
            loss.backward()
            del loss
            torch.cuda.synchronize()
            marks.append(torch.cuda.memory_allocated(dev))
        assert marks[2] == marks[7], marks
        assert max(marks[2:]) == min(marks[2:]), marks
    finally:
        gc.enable()
        _unwrap(ddp)


# ------------------------------------------------------------------ stand-alone forms
def test_bboxes_iou_golden_bit_exact(dev, golden):
    """yolo/model/yololoss.py:16-91 against the reference's own outputs (tests/golden/iou_nms.npz): bit-exact."""
    from yolov4_amd.yolo.model.yololoss import bboxes_iou
    g = golden('iou_nms')
    for tag, xyxy in (('xyxy', True), ('c', False)):
        a = torch.from_numpy(g[f'a_{tag}'].copy()).to(dev)
        b = torch.from_numpy(g[f'b_{tag}'].copy()).to(dev)
        got = bboxes_iou(a, b, xyxy=xyxy).cpu().numpy()
        assert got.dtype == np.float32 and np.array_equal(got, g[f'iou_{tag}']), tag
        assert np.array_equal(got, H.bboxes_iou(g[f'a_{tag}'], g[f'b_{tag}'], xyxy))
    with pytest.raises(IndexError):
        bboxes_iou(torch.zeros((3, 5), device=dev), torch.zeros((2, 4), device=dev))
    assert bboxes_iou(torch.zeros((0, 4), device=dev), torch.zeros((2, 4), device=dev)).shape == (0, 2)
    # degenerate / NaN boxes follow torch.max/min NaN propagation
    a = torch.tensor([[0., 0., 0., 0.], [float('nan'), 0., 1., 1.], [0., 0., 2., 2.]])
    b = torch.tensor([[0., 0., 0., 0.], [1., 1., 3., 3.]])
    ref = H.bboxes_iou(a.numpy(), b.numpy(), True)
    got = bboxes_iou(a.to(dev), b.to(dev)).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.nan_to_num(got), np.nan_to_num(ref))


def test_mish_standalone_matches_reference_formula(dev):
    """Mish()(x), darknet/darknet.py:14-20: x * tanh(softplus(x)), forward and autograd backward, any dense layout."""
    from yolov4_amd.darknet.darknet import Mish
    x = torch.cat([recipe.randn((2, 32, 9, 7), 3, 3.0).flatten(), torch.tensor([-30., -20., 0., 19.9, 20., 20.1, 50.])])
    for t in (x, recipe.randn((2, 32, 9, 7), 4, 2.0).contiguous(memory_format=torch.channels_last), recipe.randn((5,), 5)):
        xr = t.clone().double().requires_grad_(True)
        yr = xr * torch.tanh(F.softplus(xr))
        gy = recipe.randn(tuple(t.shape), 6)
        yr.backward(gy.double())
        xd = t.to(dev).requires_grad_(True)
        y = Mish()(xd)
        assert y.shape == t.shape and y.stride() == xd.stride()
        close(y, yr, 1e-6, 1e-5, scale=False)
        y.backward(gy.to(dev))
        close(xd.grad, xr.grad, 1e-6, 1e-5, scale=False)


@pytest.mark.parametrize('hw,target', [((19, 19), (38, 38)), ((38, 38), (75, 75)), ((5, 7), (13, 9)), ((4, 6), (12, 12)),
                                       ((9, 9), (9, 9)), ((10, 10), (7, 5))])
def test_upsample_general_nearest(dev, hw, target):
    """Upsample.forward for any target (yolov4.py:82-90): train = F.interpolate(size=target, mode='nearest')
    (e.g. S = 600: 38 -> 75); eval = integer-factor expand, which the reference can only .view() when the target
    is a multiple of the input."""
    from yolov4_amd.yolo.model.yolov4 import Upsample
    x = recipe.randn((2, 8) + hw, 11)
    up = Upsample().train()
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=target, mode='nearest')
    gy = recipe.randn((2, 8) + target, 12)
    yr.backward(gy)
    xd = x.to(dev).requires_grad_(True)
    y = up(xd, (2, 8) + target)
    assert torch.equal(y.cpu(), yr.detach())
    y.backward(gy.to(dev))
    close(xd.grad, xr.grad, 1e-6, 1e-6)
    up.eval()
    if target[0] % hw[0] == 0 and target[1] % hw[1] == 0:
        fh, fw = target[0] // hw[0], target[1] // hw[1]
        ref = x.view(2, 8, hw[0], 1, hw[1], 1).expand(2, 8, hw[0], fh, hw[1], fw).contiguous().view(2, 8, *target)
        with torch.no_grad():
            assert torch.equal(up(x.to(dev), (2, 8) + target).cpu(), ref)
    else:
        with pytest.raises(RuntimeError):
            up(x.to(dev), (2, 8) + target)


# ------------------------------------------------------------------ BASELINE configs[0] and configs[4]
# stated tolerance of the bf16 mixed-precision mode (configs[4]) against the fp32-grade mode, whole model @128, B = 2
BF16_TOL = {'score_mean': 1e-2, 'score_max': 0.2, 'box_mean': 1e-2, 'box_max': 0.4, 'loss_rel': 2e-2,
            'grad_final_head_weight': 0.3, 'grad_final_head_bias': 2e-2}
def test_config0_eval_forward_416_vs_oracle(dev):
    """BASELINE configs[0]: yolov4_default.cfg forward on 1x3x416x416 (val.py path) -> [1, 10647, 85], HIP vs the
    oracle (torch CPU fp32 restatement of the reference) on the same seeded input, SURVEY 8(c) recipe weights with
    BatchNorm statistics calibrated at 416."""
    from yolov4_amd.yolo.util.utils import postprocess
    m = _model(dev, 77)
    x_cal = recipe.randn((2, 3, 416, 416), 31)
    recipe.calibrate_bn_(m, x_cal.to(dev))
    sd = NW.empty_state_dict()
    recipe.fill_state_dict_(sd, 77)
    net = NW.RefNet(sd, CFG)
    net.calibrate(x_cal)
    torch.manual_seed(0)
    x = torch.randn(1, 3, 416, 416)
    m.eval()
    with torch.no_grad():
        out = m(x.to(dev))
    ref = net.forward_eval(x)
    assert tuple(out.shape) == (1, 10647, 85) and ref.shape == (1, 10647, 85)
    close(out[..., 4:], ref[..., 4:], 1e-4, 1e-4, scale=False)          # obj / cls: abs 1e-4 (north_star)
    close(out[..., :2], ref[..., :2], 1e-4, 1e-4)                       # centres in px
    close(out[..., 2:4], ref[..., 2:4], 1e-4, 1e-4)                     # w, h: rel 1e-4 of the box scale
    # survivor sets at a threshold that keeps ~300 candidates clear of rounding
    sc = (out[0, :, 4:5] * out[0, :, 5:]).flatten()
    thr = float(torch.sort(sc, descending=True).values[300])
    if int(((sc - thr).abs() < 1e-5 * max(thr, 1e-6)).sum()) <= 1:
        da = postprocess(out.clone(), 80, thr, 0.45)[0]
        db = H.postprocess(ref.copy(), 80, thr, 0.45)[0]
        assert da.shape == db.shape and np.array_equal(da[:, 6].cpu().numpy(), db[:, 6])


def test_whole_model_bf16_mode_tolerance(dev):
    """BASELINE configs[4] arithmetic (bf16 MFMA operands, fp32 accumulate / BN / loss / NMS) through the WHOLE
    detector, against the fp32-grade mode of the same build.  Mixed precision: operand error 2^-9 per conv.  The
    backward of a random-weight 110-layer BatchNorm network amplifies perturbations by ~1e5 (an fp32 rounding already
    moves deep gradients by percents, test_gradients_within_reference_rounding), so gradients are only comparable
    where the chain is short: the three final head convs (stated: weights within 30 %, biases within 2 % of their norm;
    the weight gradients also carry the few-percent error of the activations that reach the heads).  Stated tolerance for the
    forward results (BF16_TOL): mean / max of the eval score and box differences, and the training loss."""
    import yolov4_amd
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    m = _model(dev, 77)
    x_cal = recipe.randn((8, 3, 128, 128), 31).to(dev)
    x = recipe.randn((8, 3, 128, 128), 32).to(dev)
    labels = recipe.synth_labels(8, 128, 33)
    crit = YOLOLoss(CFG, 0.7, device=dev, mutate_outputs=False)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    res = {}
    old = yolov4_amd.get_conv_mode()
    try:
        for mode in ('bf16x3', 'bf16'):
            yolov4_amd.set_conv_mode(mode)
            m.load_state_dict(sd0)
            recipe.calibrate_bn_(m, x_cal)
            m.eval()
            with torch.no_grad():
                ev = m(x).clone()
            m.load_state_dict(sd0)
            m.train()
            m.zero_grad(set_to_none=True)
            loss = crit(m(x), {'padded_labels': labels})
            loss.backward()
            res[mode] = (ev, float(loss), {k: p.grad.double().clone() for k, p in m.named_parameters()})
    finally:
        yolov4_amd.set_conv_mode(old)
    (e0, l0, g0), (e1, l1, g1) = res['bf16x3'], res['bf16']
    assert torch.isfinite(e1).all()
    ds = (e1[..., 4:] - e0[..., 4:]).abs()
    db = (e1[..., :4] - e0[..., :4]).abs() / e0[..., :4].abs().max()
    rel = {k: float((g1[k] - g0[k]).norm()) / max(float(g0[k].norm()), 1e-12) for k in g0}
    final = {k: v for k, v in rel.items() if k.startswith(('head.yolo1.1.', 'head.yolo2.1.', 'head.yolo3.1.'))}
    order = [k for k, _ in m.named_parameters()]
    print(f'bf16 vs fp32-grade: score diff mean {float(ds.mean()):.3e} max {float(ds.max()):.3e}; box diff (of scale) mean '
          f'{float(db.mean()):.3e} max {float(db.max()):.3e}; loss {l0:.4f} -> {l1:.4f}; grad rel diff final head convs '
          f'{ {k: round(v, 4) for k, v in final.items()} } median {float(np.median(list(rel.values()))):.3e}; by depth '
          f'{[round(rel[k], 3) for k in order[::-1][::24]]}')
    assert float(ds.mean()) <= BF16_TOL['score_mean'] and float(ds.max()) <= BF16_TOL['score_max']
    assert float(db.mean()) <= BF16_TOL['box_mean'] and float(db.max()) <= BF16_TOL['box_max']
    assert abs(l1 - l0) <= BF16_TOL['loss_rel'] * abs(l0), (l0, l1)
    assert max(v for k, v in final.items() if k.endswith('weight')) <= BF16_TOL['grad_final_head_weight'], final
    assert max(v for k, v in final.items() if k.endswith('bias')) <= BF16_TOL['grad_final_head_bias'], final
    assert not torch.equal(e0, e1)                                       # the mode switch really changed the arithmetic


# ------------------------------------------------------------------ stored gradients, conditioning-aware
def test_train_step_stored_gradients_within_reference_rounding(dev, golden):
    """Every gradient tensor the reference's train step left in tests/golden/model.npz (stem, stage1/3/5, neck, the
    spp slice, the heads) against the HIP path, with a per-tensor bound instead of a norm percentage: the reference's
    own fp32 arithmetic is e32[k] away from an fp64 evaluation of the same backward (ill-conditioned at B = 2 batch
    statistics), measured here on the CPU for the same head gradient; HIP-vs-reference must stay within
    4 x e32[k] + 1e-4 (both are fp32 evaluations of one truth: e_hip <= 3 e32 as in
    test_gradients_within_reference_rounding, + e32 of the fixture itself)."""
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    g = golden('model')
    seed = int(g['seed'])
    m = _model(dev, seed).train()
    m.zero_grad(set_to_none=True)
    x = recipe.randn((2, 3, 128, 128), 80)
    labels = recipe.synth_labels(2, 128, 81, counts=[9, 21])
    crit = YOLOLoss(CFG, ignore_thresh=0.7, device=dev, mutate_outputs=False)
    loss = crit(m(x.to(dev)), {'padded_labels': labels})
    loss.backward()
    named = dict(m.named_parameters())

    sd = NW.empty_state_dict()
    recipe.fill_state_dict_(sd, seed)
    net32 = NW.RefNet(sd, CFG)
    lg32 = net32.forward_train(x)
    _, G = H.yolo_loss([t.detach().numpy() for t in lg32], labels.numpy(), CFG, 0.7)
    G = [torch.from_numpy(t) for t in G]
    torch.autograd.backward(lg32, G)
    net64 = NW.RefNet({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, CFG)
    torch.autograd.backward(net64.forward_train(x.double()), [t.double() for t in G])

    checked = 0
    for k in g.files:
        if k.startswith('train128.grad.'):
            name, sl = k[14:], (slice(None),)
        elif k.startswith('train128.gradslice.'):
            name, sl = k[19:], (slice(0, 8), slice(0, 16))
        else:
            continue
        ref = torch.from_numpy(g[k]).double()
        t64 = net64.p[name].grad[sl]
        e32 = float((net32.p[name].grad.double()[sl] - t64).norm() / t64.norm())
        got = named[name].grad.double().cpu()[sl]
        err = float((got - ref).norm() / ref.norm())
        assert err <= 4.0 * e32 + 1e-4, (name, err, e32)
        checked += 1
    assert checked == 13


def test_permutation_invariance_proves_its_excuse(dev):
    """BASELINE configs[2] size (608x608, bs = 64, train): permute the batch.  Loss, batch statistics and gradients
    are sums over the batch, so only the fp32 summation order changes.  The backward of a random-weight 110-layer
    BatchNorm network amplifies such rounding-level changes enormously, so the claim is PROVEN rather than asserted:
      (1) the same order twice gives the same bits (the kernels are deterministic);
      (2) the only discrete function of the forward pass, the ignore mask (IoU(pred, truth) > 0.7,
          yololoss.py:281-300), is compared cell by cell between the two orders;
      (3) a third run in the ORIGINAL order with the input perturbed by one fp32 ulp measures what a rounding-level
          change does to each gradient; the permutation may move a gradient at most 4x as far (+1e-5).
    With no mask flip the loss agrees to 1e-5 and the forward logits to 1e-4 of their range."""
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    m = _model(dev, 4321).train()                        # SURVEY 8(c) recipe weights: well-conditioned activations
    crit = YOLOLoss(CFG, ignore_thresh=0.7, device=dev, mutate_outputs=False)
    B = 64
    x = recipe.randn((B, 3, 608, 608), 500).to(dev)
    labels = recipe.synth_labels(B, 608, 501)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(9))
    inv = torch.argsort(perm).to(dev)
    sign = (recipe.rand((B, 3, 608, 608), 502) < 0.5).to(dev)
    x_ulp = torch.where(sign, x * (1.0 + 2.0 ** -23), x * (1.0 - 2.0 ** -23))

    def run(xx, ll):
        m.zero_grad(set_to_none=True)
        outs = m(xx)
        logits = [o['output'].detach().clone() for o in outs]
        loss = crit(outs, {'padded_labels': ll})
        loss.backward()
        masks = [crit.last[l]['obj_mask'].clone() for l in range(3)]
        return float(loss), {k: p.grad.clone() for k, p in m.named_parameters()}, masks, logits

    l0, g0, m0, o0 = run(x, labels)
    l0b, g0b, _, _ = run(x, labels)
    assert l0 == l0b and all(torch.equal(g0[k], g0b[k]) for k in g0), 'same order, same input: different bits'
    del g0b
    l1, g1, m1, o1 = run(x[perm.to(dev)], labels[perm])
    l2, g2, m2, _ = run(x_ulp, labels)
    flips = sum(int((a != b[inv]).sum()) for a, b in zip(m0, m1))
    flips_ulp = sum(int((a != b).sum()) for a, b in zip(m0, m2))
    fwd = max(float((a - b[inv]).abs().max()) for a, b in zip(o0, o1))
    rel = lambda ga, gb, k: float((ga[k].double() - gb[k].double()).norm()) / max(float(ga[k].double().norm()), 1e-12)
    e_perm = {k: rel(g0, g1, k) for k in g0}
    e_ulp = {k: rel(g0, g2, k) for k in g0}
    ratio = {k: e_perm[k] / (e_ulp[k] + 2.5e-6) for k in g0}
    worst = sorted(ratio.items(), key=lambda kv: -kv[1])[:3]
    head = max(v for k, v in e_perm.items() if k.startswith('head.'))
    print(f'permutation: {flips} (ulp run: {flips_ulp}) of {sum(a.numel() for a in m0)} obj_mask cells differ; max |dlogit| {fwd:.2e}; '
          f'rel. gradient difference: permuted head {head:.3e} worst {max(e_perm.values()):.3e}; one-ulp input head '
          f'{max(v for k, v in e_ulp.items() if k.startswith("head.")):.3e} worst {max(e_ulp.values()):.3e}; '
          f'worst ratios {[(k, round(v, 2)) for k, v in worst]}')
    assert l0 == l0 and flips <= 4, flips
    if flips == 0:
        assert abs(l0 - l1) <= 1e-5 * abs(l0), (l0, l1)
        assert fwd <= 1e-4
    if flips == 0 and flips_ulp == 0:
        for k in g0:
            assert e_perm[k] <= 4.0 * e_ulp[k] + 1e-5, (k, e_perm[k], e_ulp[k])


# ------------------------------------------------------------------ optimizers
def _opt_params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 32, 3, 3), (255,), (128,), (32, 3, 3, 3), (1024, 512, 1, 1), (7,), (300, 130, 3, 3)]
    ref_p = [torch.randn(s, generator=g) for s in shapes]
    hip_p = [torch.nn.Parameter((p.clone().contiguous(memory_format=torch.channels_last) if p.dim() == 4 else p.clone()).to(dev))
             for p in ref_p]
    return g, [torch.nn.Parameter(p) for p in ref_p], hip_p


def test_multi_tensor_adam_bit_equal_to_per_tensor_kernel_and_one_launch(dev):
    """y4_adam_multi_step_f32 (one launch for all tensors, 64 Ki-element chunks incl. ragged tails and unaligned
    slots) must reproduce y4_adam_step_f32 bit for bit, and torch.optim.Adam to rounding."""
    from yolov4_amd import ops
    from yolov4_amd._lib import check, lib
    from yolov4_amd.yolo.optim.optimizers.build import FusedAdam
    g, ref_p, hip_p = _opt_params(dev)
    # an unaligned parameter: a view starting 4 bytes into a buffer
    base = torch.randn(1001, generator=g).to(dev)
    odd = torch.nn.Parameter(base[1:])
    ref_odd = torch.nn.Parameter(base[1:].cpu().clone())
    hip_p.append(odd); ref_p.append(ref_odd)
    one_p = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in hip_p]
    one_m = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in one_p]
    one_v = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in one_p]
    ref = torch.optim.Adam([{'params': ref_p[:4]}, {'params': ref_p[4:], 'weight_decay': 0.01}], lr=3e-4, betas=(0.9, 0.999), eps=1e-8)
    hip = FusedAdam([{'params': hip_p[:4]}, {'params': hip_p[4:], 'weight_decay': 0.01}], lr=3e-4)
    for step in range(1, 5):
        for i, (rp, hp) in enumerate(zip(ref_p, hip_p)):
            gr = torch.randn(rp.shape, generator=g) * (10.0 ** (step - 3))
            rp.grad = gr.clone()
            hp.grad = (gr.contiguous(memory_format=torch.channels_last) if gr.dim() == 4 else gr).to(dev)
            wd = 0.0 if i < 4 else 0.01
            check(lib().y4_adam_step_f32(ops._ptr(one_p[i]), ops._ptr(hp.grad), ops._ptr(one_m[i]), ops._ptr(one_v[i]),
                                         one_p[i].numel(), 3e-4, 0.9, 0.999, 1e-8, wd, step, 1.0, ops._stream()))
        ref.step(); hip.step()
        for a, b in zip(hip_p, one_p):
            assert torch.equal(a.detach(), b.detach())
    assert hip.launches == 4                                             # ONE launch per step
    for rp, hp in zip(ref_p, hip_p):
        np.testing.assert_allclose(hp.detach().cpu().numpy(), rp.detach().numpy(), rtol=2e-6, atol=2e-7)


def test_fused_sgd_matches_torch_sgd(dev):
    """build_optimizer's SGD branch (yolo/optim/optimizers/build.py:25-28, sgd.py:14-15)."""
    from yolov4_amd.yolo.optim.optimizers.build import FusedSGD
    g, ref_p, hip_p = _opt_params(dev, 3)
    ref = torch.optim.SGD([{'params': ref_p[:3]}, {'params': ref_p[3:], 'weight_decay': 0.0}], lr=0.05, momentum=0.9, weight_decay=1e-3)
    hip = FusedSGD([{'params': hip_p[:3]}, {'params': hip_p[3:], 'weight_decay': 0.0}], lr=0.05, momentum=0.9, weight_decay=1e-3)
    for step in range(4):
        for rp, hp in zip(ref_p, hip_p):
            gr = torch.randn(rp.shape, generator=g)
            rp.grad = gr.clone()
            hp.grad = (gr.contiguous(memory_format=torch.channels_last) if gr.dim() == 4 else gr).to(dev)
        if step == 2:
            for o in (ref, hip):
                o.param_groups[0]['lr'] = 0.01
        ref.step(); hip.step()
    for rp, hp in zip(ref_p, hip_p):
        np.testing.assert_allclose(hp.detach().cpu().numpy(), rp.detach().numpy(), rtol=2e-6, atol=2e-6)


def test_adam_state_restored_from_reference_checkpoint_layout(dev):
    """torch.optim.Adam state from a reference checkpoint is contiguous OIHW while the
    parameters here are KRSC; load_state_dict keeps strides, so the step must re-lay the moments, not pair them by
    raw offset."""
    from yolov4_amd.yolo.optim.optimizers.build import FusedAdam
    g = torch.Generator().manual_seed(5)
    w = torch.randn((16, 8, 3, 3), generator=g)
    ref_p = torch.nn.Parameter(w.clone())
    ref = torch.optim.Adam([ref_p], lr=1e-2)
    for _ in range(2):
        ref_p.grad = torch.randn(w.shape, generator=g)
        ref.step()
    state = copy.deepcopy(ref.state_dict())                              # as torch.load hands it: contiguous OIHW moments, tensor `step`
    hip_p = torch.nn.Parameter(ref_p.detach().clone().contiguous(memory_format=torch.channels_last).to(dev))
    hip = FusedAdam([hip_p], lr=1e-2)
    hip.load_state_dict(state)
    gr = torch.randn(w.shape, generator=g)
    ref_p.grad = gr.clone(); hip_p.grad = gr.to(dev)                      # contiguous gradient as well
    ref.step(); hip.step()
    np.testing.assert_allclose(hip_p.detach().cpu().numpy(), ref_p.detach().numpy(), rtol=2e-6, atol=2e-7)
    assert hip.state[hip_p]['exp_avg'].stride() == hip_p.stride()


# ------------------------------------------------------------------ drop-in DDP loop + harness
def test_reference_loop_through_ddp_and_optimizer_zero_grad(dev):
    """the reference's loop calls optimizer.zero_grad() (build.py:53,67-69).  torch's default
    sets every p.grad to None; the bucket slots must heal, every gradient must live inside its flat bucket when the
    exchange is issued, the conv kernels must write filter gradients in place (no temporaries), and the result must
    equal plain autograd."""
    from yolov4_amd import ops
    from yolov4_amd.ddp import BucketedDDP
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    m = _model(dev, 99).train()
    x = recipe.randn((2, 3, 128, 128), 80).to(dev)
    labels = recipe.synth_labels(2, 128, 81, counts=[9, 21])
    crit = YOLOLoss(CFG, 0.7, device=dev, mutate_outputs=False)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.zero_grad(set_to_none=True)
    crit(m(x), {'padded_labels': labels}).backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.load_state_dict(sd)
    m.zero_grad(set_to_none=True)
    ddp = BucketedDDP(m)                                                 # the one-line swap of main_amp.py:131
    try:
        torch_opt = torch.optim.SGD(m.parameters(), lr=0.0)
        for variant in ('set_to_none', 'in_place', 'module_zero_grad'):
            if variant == 'set_to_none':
                torch_opt.zero_grad()                                    # torch default: p.grad = None everywhere
                assert all(p.grad is None for p in m.parameters())
            elif variant == 'in_place':
                torch_opt.zero_grad(set_to_none=False)
            else:
                ddp.zero_grad()
            m.load_state_dict(sd)
            ops.WGRAD_STATS['in_place'] = ops.WGRAD_STATS['temporary'] = 0
            crit(ddp(x), {'padded_labels': labels}).backward()           # no finish_backward(): autograd callback
            assert ddp._finished and all(b.pending == 0 and b.launched for b in ddp.buckets), \
                (variant, ddp._finished, [(b.pending, b.launched, len(b.params)) for b in ddp.buckets])
            for b in ddp.buckets:
                lo, hi = b.flat.data_ptr(), b.flat.data_ptr() + b.flat.numel() * 4
                for p in b.params:
                    assert lo <= p.grad.data_ptr() < hi, variant
                    assert p.grad.data_ptr() % 16 == 0
            for k, p in m.named_parameters():
                assert torch.equal(p.grad, ref[k]), (variant, k)
            if variant != 'in_place':                                    # zeroed-in-place grads are not known to be zero
                assert ops.WGRAD_STATS['temporary'] == 0 and ops.WGRAD_STATS['in_place'] == 110, ops.WGRAD_STATS
        assert ddp.stats['copied_in'] == 0
    finally:
        _unwrap(ddp)


def test_engine_train_and_validate_harness(dev):
    """yolo/engine/build.py:41-107 / :111-190 call sequence packaged: warm-up LR, loss / ACCUMULATION_STEPS, optimizer
    step every k micro-batches (exchange only on the last one), validate = eval forward -> postprocess -> COCO records."""
    from yolov4_amd.ddp import BucketedDDP
    from yolov4_amd.yolo.engine.build import train, validate
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    from yolov4_amd.yolo.optim.optimizers.build import build_optimizer
    from yolov4_amd.yolo.util.utils import COCO_CLASS_IDS
    m = _model(dev, 7).train()
    ddp = BucketedDDP(m)
    try:
        opt = build_optimizer(OPT_CFG, m)
        crit = YOLOLoss(CFG, 0.7, device=dev)
        x = recipe.randn((2, 3, 96, 96), 1)
        labels = recipe.synth_labels(2, 96, 2, counts=[4, 6])
        loader = [(x, {'padded_labels': labels})] * 4
        import argparse
        w0 = m.head.yolo1[1].conv.weight.detach().clone()
        logs = []
        train(argparse.Namespace(print_freq=2, distributed=False, world_size=1), OPT_CFG, loader, ddp, crit, opt,
              device=dev, epoch=0, log=logs.append)
        assert opt.launches == 2                                         # 4 micro-batches, ACCUMULATION_STEPS = 2
        assert math.isclose(opt.param_groups[0]['lr'], 3e-4 * 4 / (5 * 4))  # warm-up value of the last micro-batch
        assert not torch.equal(w0, m.head.yolo1[1].conv.weight.detach()) and len(logs) == 2
        assert all(float(p.grad.abs().max()) == 0.0 for p in list(m.parameters())[:3])   # zero_grad after the step

        class DS:
            class_ids = COCO_CLASS_IDS
        class Loader(list):
            dataset = DS()
        recipe.calibrate_bn_(m, recipe.randn((4, 3, 96, 96), 3).to(dev))
        imgs = recipe.rand((2, 3, 96, 96), 4)
        vl = Loader([(imgs, {'img_info': torch.tensor([[480., 640., 96., 96., 17., 0.], [333., 500., 96., 96., 42., 0.]])})])
        got = validate(vl, ddp, 0.2, 0.4, device=dev, evaluator=lambda recs, ids: (len(recs), sorted(set(ids))))
        recs = validate.records
        assert got == (len(recs), [17, 42]) or (got == (0, 0) and not recs)
        for r in recs:
            assert r['image_id'] in (17, 42) and r['category_id'] in COCO_CLASS_IDS and len(r['bbox']) == 4
            assert 0.2 <= r['score'] <= 1.0
    finally:
        _unwrap(ddp)


# ------------------------------------------------------------------ inference caches (filter planes, BN fold)
def test_inference_filter_cache_is_exact_and_expires_with_the_weights(dev):
    """Under no_grad an eval-mode ConvBNAct keeps its filter planes in a per-parameter buffer that is refreshed on every
    call and re-split only when the filter's bits changed (device-side checksum, y4_conv2d_prepare_filter_f32): results
    must be bit-identical to the per-call path whatever rewrote the weights -- a fused optimizer step (raw pointers), a
    training forward (running statistics), or writes through `.data`, which torch's version counters do not see."""
    from yolov4_amd.darknet.darknet import ConvBNAct
    from yolov4_amd.yolo.optim.optimizers.build import FusedAdam
    torch.manual_seed(3)
    m = ConvBNAct(64, 128, 3, 1, act='mish').to(dev)
    x = torch.randn(2, 64, 20, 20, device=dev).contiguous(memory_format=torch.channels_last)

    def uncached():
        m.eval()
        os.environ['Y4_NO_INFER_CACHE'] = '1'          # the per-call path (amax + split inside the call), same fused epilogue
        try:
            with torch.no_grad():
                return m(x).clone()
        finally:
            del os.environ['Y4_NO_INFER_CACHE']

    def cached():
        m.eval()
        with torch.no_grad():
            return m(x).clone()
    a, b = uncached(), cached()
    assert getattr(m.conv.weight, '_y4_prepared', None) is not None, 'the prepared-filter path did not run'
    assert torch.equal(a, b)
    assert torch.equal(cached(), b)                    # second call: planes kept, nothing re-split
    # a training forward rewrites running_mean / running_var through raw pointers
    m.train()
    m(x)
    assert torch.equal(uncached(), cached())
    # a fused optimizer step rewrites the filter through raw pointers
    opt = FusedAdam([p for p in m.parameters()], lr=1e-2)
    m.train()
    m(x).square().mean().backward()
    before = cached()
    opt.step()
    after_u, after_c = uncached(), cached()
    assert torch.equal(after_u, after_c)
    assert not torch.equal(before, after_c)
    # writes through .data leave _version alone: filter, BatchNorm affine and running statistics
    v0 = m.conv.weight._version
    m.conv.weight.data.mul_(1.5)
    assert m.conv.weight._version == v0
    c1 = cached()
    assert torch.equal(uncached(), c1) and not torch.equal(c1, after_c)
    m.norm.running_mean.data.add_(0.25)
    m.norm.weight.data.mul_(0.5)
    c2 = cached()
    assert torch.equal(uncached(), c2) and not torch.equal(c2, c1)
    # one element poked: still seen (the fingerprint is an exact checksum, not a sample)
    m.conv.weight.data[77, 13, 1, 2] += 1.0
    c3 = cached()
    assert torch.equal(uncached(), c3) and not torch.equal(c3, c2)


def test_eval_forward_issues_no_host_sync(dev):
    """a Python truth test on a device tensor (`cell or new_cell`) is a hidden host-device sync.  The whole
    no_grad eval forward must enqueue without one (decode included; postprocess is what reads results back)."""
    m = _model(dev)
    x = recipe.randn((2, 3, 128, 128), 9).to(dev)
    recipe.calibrate_bn_(m, recipe.randn((8, 3, 128, 128), 10).to(dev))
    m.eval()
    with torch.no_grad():
        m(x)                                           # warm-up: allocations, prepared buffers
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode('error')
        try:
            out = m(x)
        finally:
            torch.cuda.set_sync_debug_mode('default')
    assert out.shape[0] == 2 and bool(torch.isfinite(out).all())


def test_amax_cell_outliving_its_ring_half_is_not_trusted(dev):
    """operand-maximum cells come from a ring that re-zeroes a half on re-entry.  A tensor kept across
    a full turn of the ring (8192 cell allocations; a half is re-zeroed on re-entry) must not read another tensor's maximum: its tag reads as expired and the consumer
    takes the maximum again."""
    from yolov4_amd import ops
    from yolov4_amd.darknet.darknet import ConvBNAct
    torch.manual_seed(5)
    a = ConvBNAct(32, 64, 1, 1, act='mish').to(dev).train()
    b = ConvBNAct(64, 64, 3, 1, act='leaky_relu').to(dev).train()
    x = (1e3 * torch.randn(2, 32, 24, 24, device=dev)).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        z = a(x)
        cell = z.y4_amax
        assert ops.live(cell) is cell
        want = b(z).clone()                            # fresh tag
        for _ in range(9000):                          # the ring (8192 words) comes round to the cell's half again
            ops.new_amax(dev)
        assert ops.live(cell) is None and ops.amax_of(z) is None
        # the recycled word now holds somebody else's (tiny) maximum: trusting it would overflow fp16 -> inf / NaN
        cell.fill_(torch.tensor(1e-6).view(torch.int32).item())
        got = b(z)
    assert bool(torch.isfinite(got).all())
    assert torch.equal(got, want)


def test_graft_entry_smoke_runs(dev):
    """The driver's smoke() (one tiny train step + eval + NMS against the oracle) must pass on this build."""
    import __graft_entry__ as g
    g.smoke()
