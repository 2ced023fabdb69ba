# -*- coding: utf-8 -*-
"""Pins oracle/ (the CPU restatement) against the fixtures produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import recipe
from oracle import head as H
from oracle import network as NW

CFG = recipe.MODEL_CFG


def close(a, b, atol=1e-5, rtol=1e-5):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), atol=atol, rtol=rtol)


def test_bboxes_iou(golden):
    g = golden('iou_nms')
    close(H.bboxes_iou(g['a_xyxy'], g['b_xyxy'], True), g['iou_xyxy'], 1e-6, 1e-5)
    close(H.bboxes_iou(g['a_c'], g['b_c'], False), g['iou_c'], 1e-6, 1e-5)
    with pytest.raises(IndexError):
        H.bboxes_iou(np.zeros((2, 3), np.float32), np.zeros((2, 4), np.float32))


def test_nms(golden):
    g = golden('iou_nms')
    assert np.array_equal(H.nms(g['box'], 0.45, score=g['score']), g['keep45'])
    assert np.array_equal(H.nms(g['box'], 0.3, score=g['score'], limit=7), g['keep30_lim'])
    assert np.array_equal(H.nms(g['box'], 0.5), g['keep_noscore'])
    k = H.nms(np.zeros((0, 4), np.float32), 0.5, score=np.zeros((0,), np.float32))
    assert k.shape == (0,) and k.dtype == np.int32
    assert H.nms(g['box'], 0.45, score=g['score']).dtype == np.int32


def test_nms_tie_order_is_defined():
    box = np.array([[0, 0, 10, 10], [100, 100, 110, 110], [0, 0, 10, 10]], np.float32)
    sc = np.array([0.5, 0.5, 0.5], np.float32)
    assert H.nms(box, 0.5, score=sc).tolist() == [0, 1]


@pytest.mark.parametrize('l', [0, 1, 2])
def test_yolo_decode(golden, l):
    g = golden('yololayer')
    out, pred = H.yolo_decode(g[f'x{l}'], l, CFG, True)
    close(out, g[f'train_output{l}'], 1e-6, 1e-6)
    close(pred, g[f'train_pred{l}'], 1e-5, 1e-6)
    close(H.yolo_decode(g[f'x{l}'], l, CFG, False), g[f'eval_out{l}'], 1e-5, 1e-6)
    go = recipe.randn(tuple(out.shape), 200 + l).numpy()
    gp = recipe.randn(tuple(pred.shape), 300 + l).numpy()
    close(H.yolo_decode_backward(g[f'x{l}'], go, gp, l, CFG), g[f'grad_x{l}'], 1e-5, 1e-5)


@pytest.mark.parametrize('l', [0, 1, 2])
def test_build_target_bit_exact(golden, l):
    g = golden('yololoss')
    out, pred = H.yolo_decode(g[f'logits{l}'], l, CFG, True)
    tgt, obj, tm, ts = H.build_target(out, pred, l, g['labels'], CFG, 0.7)
    assert np.array_equal(obj, g[f'obj_mask{l}'])              # ignore mask + positives: exact
    assert np.array_equal(tm[..., 0], g[f'tgt_mask{l}'])
    assert (tm == tm[..., :1]).all()
    assert np.array_equal(tgt != 0, g[f'target{l}'] != 0)      # anchor/cell/class indices: exact
    close(tgt, g[f'target{l}'], 1e-6, 1e-6)
    close(ts, g[f'tgt_scale{l}'], 1e-6, 1e-6)


def test_yolo_loss(golden):
    g = golden('yololoss')
    logits = [g[f'logits{l}'] for l in range(3)]
    for l in range(3):
        out, pred = H.yolo_decode(logits[l], l, CFG, True)
        r = H.yolo_loss_layer(out, pred, l, g['labels'], CFG, 0.7)
        assert abs(r['loss'] - float(g[f'loss_layer{l}'])) <= 1e-5 * abs(float(g[f'loss_layer{l}']))
        close(r['mutated_output'], g[f'mutated_output{l}'], 1e-6, 1e-6)
    loss, grads = H.yolo_loss(logits, g['labels'], CFG, 0.7)
    assert abs(loss - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    for l in range(3):
        close(grads[l], g[f'grad_logits{l}'], 1e-5, 1e-4)


def test_postprocess(golden):
    g = golden('postprocess')
    for case in range(2):
        conf, thre = g[f'params{case}']
        p = g[f'pred{case}'].copy()
        out = H.postprocess(p, 80, conf, thre)
        close(p[:, :, :4], g[f'xyxy{case}'], 0, 0)            # in-place xyxy side effect, exact
        for b in range(len(p)):
            if bool(g[f'isnone{case}_{b}']):
                assert out[b] is None
            else:
                assert out[b].shape == g[f'det{case}_{b}'].shape
                assert np.array_equal(out[b], g[f'det{case}_{b}'])   # survivor set and order: exact


def _load_cba(g, name):
    cin, cout, k, s, bn, bias, B, Hh = [int(v) for v in g[f'{name}.cfg']]
    act = str(g[f'{name}.act'])
    sd = {kk[len(name) + 4:]: torch.from_numpy(g[kk].copy()) for kk in g.files if kk.startswith(name + '.sd.')}
    return cin, cout, k, s, bool(bn), act, sd


def test_convbnact(golden):
    g = golden('convbnact')
    for name in [str(n) for n in g['names']]:
        cin, cout, k, s, bn, act, sd = _load_cba(g, name)
        net = NW.RefNet({'m.' + kk: v for kk, v in sd.items()})
        x = torch.from_numpy(g[f'{name}.x'].copy())
        net.training = False
        with torch.no_grad():
            close(net.cba(x, 'm', k, s, act, bn).numpy(), g[f'{name}.eval_y'], 1e-5, 1e-5)
        net.training = True
        xin = x.clone().requires_grad_(True)
        y = net.cba(xin, 'm', k, s, act, bn)
        close(y.detach().numpy(), g[f'{name}.train_y'], 1e-5, 1e-5)
        (y * torch.from_numpy(g[f'{name}.gy'].copy())).sum().backward()
        close(xin.grad.numpy(), g[f'{name}.gx'], 1e-4, 1e-4)
        for kk in g.files:
            if kk.startswith(name + '.grad.'):
                close(net.p['m.' + kk[len(name) + 6:]].grad.numpy(), g[kk], 1e-4, 1e-4)
            if kk.startswith(name + '.after.'):
                close(net.p['m.' + kk[len(name) + 7:]].numpy(), g[kk], 1e-5, 1e-5)


def _block_sd(module_keys_fn, seed):
    sd = module_keys_fn()
    recipe.fill_state_dict_(sd, seed)
    return sd


def _cba_keys(sd, pre, cin, cout, k):
    sd[pre + '.conv.weight'] = torch.zeros(cout, cin, k, k)
    for n in ('weight', 'bias', 'running_mean', 'running_var'):
        sd[f'{pre}.norm.{n}'] = torch.zeros(cout)
    sd[pre + '.norm.num_batches_tracked'] = torch.zeros((), dtype=torch.int64)


def test_blocks(golden):
    g = golden('blocks')

    def check(name, sd, fn, prefix='m.'):
        net = NW.RefNet(sd)
        x = torch.from_numpy(g[f'{name}.x'].copy())
        net.training = True
        xin = x.clone().requires_grad_(True)
        y = fn(net, xin)
        close(y.detach().numpy(), g[f'{name}.train_y'], 1e-4, 1e-4)
        gy = recipe.randn(tuple(y.shape), {'resblock': 602, 'csp0': 603, 'csp': 604, 'spp': 605}[name])
        (y * gy).sum().backward()
        close(xin.grad.numpy(), g[f'{name}.gx'], 2e-4, 1e-3)
        keys = [str(q) for q in g[f'{name}.gradnorm_keys']]
        for kk, ref in zip(keys, g[f'{name}.gradnorm']):
            got = float(net.p[prefix + kk].grad.double().norm())
            assert abs(got - ref) <= 1e-3 * max(ref, 1e-3), (name, kk, got, ref)
        net.training = False          # golden eval ran after the train forward (running stats updated once)
        with torch.no_grad():
            close(fn(net, x).numpy(), g[f'{name}.eval_y'], 1e-4, 1e-4)

    sd = {}
    for i in range(2):
        _cba_keys(sd, f'm.module_list.{i}.0', 32, 32, 1); _cba_keys(sd, f'm.module_list.{i}.1', 32, 32, 3)
    recipe.fill_state_dict_({k[2:]: v for k, v in sd.items()}, 601)
    check('resblock', sd, lambda n, x: n.resblock(x, 'm', 2))

    sd = {}
    _cba_keys(sd, 'm.base', 32, 64, 3); _cba_keys(sd, 'm.part1', 64, 64, 1); _cba_keys(sd, 'm.part2_1_1', 64, 64, 1)
    _cba_keys(sd, 'm.part2_1_2.0', 64, 32, 1); _cba_keys(sd, 'm.part2_1_2.1', 32, 64, 3)
    _cba_keys(sd, 'm.part2_2', 64, 64, 1); _cba_keys(sd, 'm.transition', 128, 64, 1)
    recipe.fill_state_dict_({k[2:]: v for k, v in sd.items()}, 602)
    check('csp0', sd, lambda n, x: n.csp0(x, 'm'))

    sd = {}
    _cba_keys(sd, 'm.base', 32, 64, 3); _cba_keys(sd, 'm.part1', 64, 32, 1); _cba_keys(sd, 'm.part2.0', 64, 32, 1)
    for i in range(2):
        _cba_keys(sd, f'm.part2.1.module_list.{i}.0', 32, 32, 1); _cba_keys(sd, f'm.part2.1.module_list.{i}.1', 32, 32, 3)
    _cba_keys(sd, 'm.part2.2', 32, 32, 1); _cba_keys(sd, 'm.transition', 64, 64, 1)
    recipe.fill_state_dict_({k[2:]: v for k, v in sd.items()}, 603)
    check('csp', sd, lambda n, x: n.csp(x, 'm', 2))

    sd = {}
    _cba_keys(sd, 'neck.spp.conv1.0', 1024, 512, 1); _cba_keys(sd, 'neck.spp.conv1.1', 512, 1024, 3)
    _cba_keys(sd, 'neck.spp.conv1.2', 1024, 512, 1); _cba_keys(sd, 'neck.spp.conv2', 2048, 512, 1)
    recipe.fill_state_dict_({k[len('neck.spp.'):]: v for k, v in sd.items()}, 604)
    check('spp', sd, lambda n, x: n.spp(x), prefix='neck.spp.')

    up_x = torch.from_numpy(g['up.x'].copy())
    close(NW.RefNet.up(up_x, (6, 6)).numpy(), g['up.train'], 0, 0)
    close(NW.RefNet.up(up_x, (6, 6)).numpy(), g['up.eval'], 0, 0)


def test_state_dict_spec_matches_reference(golden):
    g = golden('model')
    spec = NW.yolov4_state_dict_spec()
    assert [k for k, _ in spec] == [str(k) for k in g['keys']]
    assert [str(tuple(s)) for _, s in spec] == [str(s) for s in g['shapes']]
    assert len(spec) == 648


@pytest.fixture(scope='module')
def recipe_sd(golden):
    sd = NW.empty_state_dict()
    recipe.fill_state_dict_(sd, int(golden('model')['seed']))
    return sd


def test_model_eval_forward(golden, recipe_sd):
    g = golden('model')
    net = NW.RefNet(recipe_sd, CFG)
    net.calibrate(recipe.randn((8, 3, 64, 64), 77))
    for k in g.files:
        if k.startswith('cal.'):
            close(net.p[k[4:]].numpy(), g[k], 1e-4, 1e-3)
    out = net.forward_eval(recipe.randn((2, 3, 64, 64), 78))
    ref = g['eval64.out']
    close(out[..., 4:], ref[..., 4:], 1e-4, 1e-4)            # obj / cls: abs 1e-4
    close(out[..., :4], ref[..., :4], 1e-3, 1e-4)            # boxes (px, values up to hundreds): rel 1e-4
    det = H.postprocess(out.copy(), 80, 0.12, 0.4)
    for b in range(2):
        rd = g[f'eval64.det{b}']
        assert det[b].shape == rd.shape
        assert np.array_equal(det[b][:, 6], rd[:, 6])        # same classes in the same order
        close(det[b], rd, 1e-3, 1e-4)


def test_model_train_step(golden, recipe_sd):
    g = golden('model')
    net = NW.RefNet(recipe_sd, CFG)
    x = recipe.randn((2, 3, 128, 128), 80)
    labels = recipe.synth_labels(2, 128, 81, counts=[9, 21])
    loss, lg = net.train_step(x, labels.numpy())
    assert abs(loss - float(g['train128.loss'])) <= 1e-4 * float(g['train128.loss'])
    for l in range(3):
        out, pred = H.yolo_decode(lg[l].detach().numpy(), l, CFG, True)
        close(out, g[f'train128.output{l}'], 1e-4, 1e-4)
        close(pred, g[f'train128.pred{l}'], 1e-3, 1e-4)
    keys = [str(q) for q in g['train128.gradnorm_keys']]
    for kk, ref in zip(keys, g['train128.gradnorm']):
        got = float(net.p[kk].grad.double().norm())
        assert abs(got - ref) <= 2e-3 * max(ref, 1e-6), (kk, got, ref)
    for k in g.files:
        if k.startswith('train128.grad.'):
            ref = g[k]
            close(net.p[k[14:]].grad.numpy(), ref, 1e-3 * float(np.abs(ref).max()), 1e-3)
        if k.startswith('train128.after.'):
            close(net.p[k[15:]].numpy(), g[k], 1e-5, 1e-4)
