# -*- coding: utf-8 -*-
"""Golden-vector generator.  Runs ONLY in the build container, where the
reference (zjykzj/YOLOv4) is mounted read-only at /root/reference.  It imports
the reference's hot-path modules (darknet.darknet, yolo.model.*, yolo.util.utils),
feeds them the seeded inputs of tests/recipe.py on the PyTorch CPU fp32 path and
writes small .npz fixtures next to this file.  The fixtures are data (inputs and
expected outputs); no reference source is copied.  The reference never travels to
the GPU box: tests read only the .npz files.

    python tests/golden/make_golden.py            # regenerate everything

The reference publishes no golden vectors or asserting tests of its own
(SURVEY.md §4), so these files are what pins the oracle.
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/
sys.path.insert(0, '/root/reference')

import numpy as np
import torch

import recipe

from darknet.darknet import ConvBNAct, ResBlock, CSPDownSample0, CSPDownSample   # noqa: E402
from yolo.model.yolov4 import YOLOv4, SPPBlock, Upsample                         # noqa: E402
from yolo.model.yololayer import YOLOLayer                                        # noqa: E402
from yolo.model.yololoss import YOLOLoss, bboxes_iou                              # noqa: E402
from yolo.util.utils import postprocess, nms                                      # noqa: E402

CPU = torch.device('cpu')
torch.set_num_threads(8)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays')


# --------------------------------------------------------------------------- A
def gen_iou_nms():
    rng = np.random.RandomState(11)
    a = rng.uniform(0, 50, (37, 4)).astype(np.float32)
    b = rng.uniform(0, 50, (23, 4)).astype(np.float32)
    a_xyxy = a.copy(); a_xyxy[:, 2:] = a[:, :2] + rng.uniform(1, 30, (37, 2)).astype(np.float32)
    b_xyxy = b.copy(); b_xyxy[:, 2:] = b[:, :2] + rng.uniform(1, 30, (23, 2)).astype(np.float32)
    a_c = a.copy(); a_c[:, 2:] = rng.uniform(1, 30, (37, 2)).astype(np.float32)
    b_c = b.copy(); b_c[:, 2:] = rng.uniform(1, 30, (23, 2)).astype(np.float32)
    iou_xyxy = bboxes_iou(torch.from_numpy(a_xyxy), torch.from_numpy(b_xyxy), xyxy=True)
    iou_c = bboxes_iou(torch.from_numpy(a_c), torch.from_numpy(b_c), xyxy=False)

    # nms: clustered boxes, tie-free scores
    R = 400
    ctr = rng.uniform(50, 550, (12, 2))
    k = rng.randint(0, 12, R)
    c = ctr[k] + rng.normal(0, 8, (R, 2))
    wh = rng.uniform(40, 120, (12, 2))[k] * np.exp(rng.normal(0, 0.1, (R, 2)))
    box = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    score = rng.permutation(R).astype(np.float32) / R + 0.001
    assert len(np.unique(score)) == R
    keep45 = nms(box, 0.45, score=score)
    keep30_lim = nms(box, 0.3, score=score, limit=7)
    keep_noscore = nms(box, 0.5)
    keep_empty = nms(np.zeros((0, 4), np.float32), 0.5, score=np.zeros((0,), np.float32))
    save('iou_nms.npz', a_xyxy=a_xyxy, b_xyxy=b_xyxy, iou_xyxy=iou_xyxy, a_c=a_c, b_c=b_c, iou_c=iou_c,
         box=box, score=score, keep45=keep45, keep30_lim=keep30_lim, keep_noscore=keep_noscore,
         keep_empty=keep_empty)


# --------------------------------------------------------------------------- B
def gen_yololayer():
    arrs = {}
    B = 2
    for layer_no, F in enumerate((8, 4, 2)):
        x = recipe.synth_head_logits(B, F, 100 + layer_no)
        arrs[f'x{layer_no}'] = x.clone()
        lay = YOLOLayer(recipe.MODEL_CFG, layer_no, device=CPU)
        lay.train()
        xin = x.clone().requires_grad_(True)
        r = lay(xin * 1.0)     # in-place ops on a view of a leaf are illegal; feed a non-leaf
        assert r['layer_no'] == layer_no
        arrs[f'train_output{layer_no}'] = r['output'].detach().contiguous()
        arrs[f'train_pred{layer_no}'] = r['pred'].detach().contiguous()
        # backward of a fixed linear functional of (output, pred): pins d(logits)
        go = recipe.randn(tuple(r['output'].shape), 200 + layer_no)
        gp = recipe.randn(tuple(r['pred'].shape), 300 + layer_no)
        ((r['output'] * go).sum() + (r['pred'] * gp).sum()).backward()
        arrs[f'grad_x{layer_no}'] = xin.grad.clone()
        lay.eval()
        with torch.no_grad():
            arrs[f'eval_out{layer_no}'] = lay(x.clone())
    save('yololayer.npz', **arrs)


# --------------------------------------------------------------------------- C
def make_loss_case(B, S, seed):
    """Head logits + labels with the edge cases of yololoss.py:118-371:
    an image with no labels, colliding truths (same cell+anchor, different
    class: last-writer-wins xy/wh, OR-ed class bits), preds forced close to
    truths so the ignore mask (IoU > 0.7) bites on non-assigned anchors."""
    strides = (8, 16, 32)
    counts = [7, 0, 13, 60][:B] if B <= 4 else None
    labels = recipe.synth_labels(B, S, seed, counts=counts)
    # collisions: copy truth 0 of image 0 into row 1 with another class and a tiny shift
    labels[0, 1] = labels[0, 0]
    labels[0, 1, 0] += 0.25
    labels[0, 1, 4] = (labels[0, 0, 4] + 3) % 80
    # and an exact duplicate class (bit already set)
    labels[0, 2] = labels[0, 0]
    # a third of the truths large enough to be matched to the stride-16/32 anchors
    big = labels[:, 3::3, 2:4] * 3.0
    labels[:, 3::3, 2:4] = torch.where(big > 0, big.clamp(max=float(S) * 0.9), big)
    logits = []
    for l, st in enumerate(strides):
        F = S // st
        x = recipe.synth_head_logits(B, F, seed + 10 * (l + 1))
        v = x.view(B, 3, 85, F, F)
        # push some predictions onto truths -> IoU > ignore_thresh at every anchor of that cell
        for b in range(B):
            n = int((labels[b].sum(1) > 0).sum())
            for ti in range(0, n, 2):
                tx, ty, tw, th = (labels[b, ti, :4] / st).tolist()
                i, j = int(tx), int(ty)
                if i >= F or j >= F:
                    continue
                for a in range(3):
                    aw, ah = [q / st for q in recipe.ANCHORS[recipe.ANCHOR_MASK[l][a]]]
                    fx = min(max(tx - i, 0.02), 0.98)
                    fy = min(max(ty - j, 0.02), 0.98)
                    v[b, a, 0, j, i] = float(np.log(fx / (1 - fx)))
                    v[b, a, 1, j, i] = float(np.log(fy / (1 - fy)))
                    v[b, a, 2, j, i] = float(np.log(tw / aw)) + 0.05 * a
                    v[b, a, 3, j, i] = float(np.log(th / ah)) - 0.05 * a
        logits.append(x)
    return logits, labels


def gen_yololoss():
    B, S = 4, 128
    logits, labels = make_loss_case(B, S, 4242)
    crit = YOLOLoss(recipe.MODEL_CFG, ignore_thresh=0.7, device=CPU)
    arrs = {'labels': labels.clone()}
    xs, outs = [], []
    for l in range(3):
        x = logits[l].clone().requires_grad_(True)
        lay = YOLOLayer(recipe.MODEL_CFG, l, device=CPU).train()
        r = lay(x * 1.0)
        xs.append(x); outs.append(r)
        arrs[f'logits{l}'] = logits[l].clone()
        # build_target on un-mutated copies
        tgt, obj_mask, tgt_mask, tgt_scale = crit.build_target(
            r['output'].detach().clone(), r['pred'].detach().clone(), l, labels.clone())
        arrs[f'target{l}'] = tgt
        arrs[f'obj_mask{l}'] = obj_mask
        arrs[f'tgt_mask{l}'] = tgt_mask[..., 0].contiguous()
        assert bool((tgt_mask == tgt_mask[..., :1]).all())
        arrs[f'tgt_scale{l}'] = tgt_scale
    # per-layer loss values (fresh graph each, the reference mutates outputs in place)
    for l in range(3):
        lay = YOLOLayer(recipe.MODEL_CFG, l, device=CPU).train()
        r = lay(logits[l].clone())
        arrs[f'loss_layer{l}'] = crit([r], {'padded_labels': labels.clone()}).detach()
    loss = crit(outs, {'padded_labels': labels.clone()})
    loss.backward()
    arrs['loss'] = loss.detach()
    for l in range(3):
        arrs[f'grad_logits{l}'] = xs[l].grad.clone()
        arrs[f'mutated_output{l}'] = outs[l]['output'].detach().contiguous()
    # ignore mask must be non-trivial for the fixture to mean anything
    for l in range(3):
        z = int((arrs[f'obj_mask{l}'] == 0).sum())
        p = int((arrs[f'tgt_mask{l}'] == 1).sum())
        print(f'  loss layer {l}: ignored cells {z}, positive cells {p}')
    save('yololoss.npz', **arrs)


# --------------------------------------------------------------------------- D
def gen_postprocess():
    arrs = {}
    for case, (B, N, conf, thre, seed) in enumerate([(2, 600, 0.3, 0.45, 5), (3, 1200, 0.1, 0.4, 6)]):
        pred = recipe.synth_predictions(B, N, seed)
        if case == 1:
            pred[1, :, 4] = 0.0          # image with no detections -> None
        arrs[f'pred{case}'] = pred.clone()
        arrs[f'params{case}'] = np.array([conf, thre], dtype=np.float64)
        # tie-freeness inside every (image, class)
        sc = pred[:, :, 4:5] * pred[:, :, 5:]
        for b in range(B):
            for c in range(80):
                s = sc[b, :, c]
                s = s[s >= conf]
                assert len(torch.unique(s)) == len(s), 'score tie in golden input'
        p = pred.clone()
        out = postprocess(p, 80, conf_thre=conf, nms_thre=thre)
        arrs[f'xyxy{case}'] = p[:, :, :4].clone()     # in-place side effect
        for b in range(B):
            arrs[f'det{case}_{b}'] = out[b] if out[b] is not None else np.zeros((0, 7), np.float32)
            arrs[f'isnone{case}_{b}'] = np.array(out[b] is None)
            print(f'  postprocess case {case} img {b}: {0 if out[b] is None else len(out[b])} detections')
    save('postprocess.npz', **arrs)


# --------------------------------------------------------------------------- F
def gen_convbnact():
    """Single ConvBNAct layers, data stored explicitly (tiny spatial, real channel mix)."""
    cases = [
        # name, cin, cout, k, s, act, bn, bias, B, H
        ('c3s1_mish', 32, 64, 3, 1, 'mish', True, False, 2, 10),
        ('c3s2_mish', 32, 64, 3, 2, 'mish', True, False, 2, 12),
        ('c1_leaky', 64, 32, 1, 1, 'leaky_relu', True, False, 2, 9),
        ('c3s2_leaky_odd', 32, 32, 3, 2, 'leaky_relu', True, False, 3, 11),
        ('c1_head', 64, 255, 1, 1, 'linear', False, True, 2, 6),
        ('c3_head', 32, 255, 3, 1, 'linear', False, True, 2, 7),
        ('stem', 3, 32, 3, 1, 'mish', True, False, 2, 16),
        ('c1_relu', 32, 32, 1, 1, 'relu', True, False, 2, 8),
    ]
    arrs = {'names': np.array([c[0] for c in cases])}
    for ci, (name, cin, cout, k, s, act, bn, bias, B, H) in enumerate(cases):
        torch.manual_seed(1000 + ci)
        m = ConvBNAct(cin, cout, k, s, bias=bias, bn=bn, act=act)
        sd = m.state_dict()
        recipe.fill_state_dict_(sd, 2000 + ci)
        if bn:
            sd['norm.running_mean'].copy_(recipe.randn(sd['norm.running_mean'].shape, 2100 + ci, 0.2))
            sd['norm.running_var'].copy_(recipe.rand(sd['norm.running_var'].shape, 2200 + ci) + 0.5)
        m.load_state_dict(sd)
        x = recipe.randn((B, cin, H, H), 3000 + ci)
        arrs[f'{name}.cfg'] = np.array([cin, cout, k, s, int(bn), int(bias), B, H])
        arrs[f'{name}.act'] = np.array(act)
        arrs[f'{name}.x'] = x.clone()
        for kk, vv in sd.items():
            arrs[f'{name}.sd.{kk}'] = vv.clone()
        m.eval()
        with torch.no_grad():
            arrs[f'{name}.eval_y'] = m(x.clone())
        m.train()
        xin = x.clone().requires_grad_(True)
        y = m(xin)
        gy = recipe.randn(tuple(y.shape), 4000 + ci)
        (y * gy).sum().backward()
        arrs[f'{name}.gy'] = gy
        arrs[f'{name}.train_y'] = y.detach()
        arrs[f'{name}.gx'] = xin.grad.clone()
        for pn, p in m.named_parameters():
            arrs[f'{name}.grad.{pn}'] = p.grad.clone()
        for kk, vv in m.state_dict().items():
            if 'running' in kk or 'num_batches' in kk:
                arrs[f'{name}.after.{kk}'] = vv.clone()
    save('convbnact.npz', **arrs)


def gen_blocks():
    """ResBlock / CSP stages / SPP / Upsample at tiny sizes: pins wiring
    (cat order [x2, x1], SPP pools 5/9/5, residual placement)."""
    arrs = {}

    def run(name, mod, x, seed):
        sd = mod.state_dict()
        recipe.fill_state_dict_(sd, seed)
        mod.load_state_dict(sd)
        mod.train()
        xin = x.clone().requires_grad_(True)
        y = mod(xin)
        gy = recipe.randn(tuple(y.shape), seed + 1)
        (y * gy).sum().backward()
        arrs[f'{name}.x'] = x
        arrs[f'{name}.train_y'] = y.detach()
        arrs[f'{name}.gx'] = xin.grad.clone()
        gn = {pn: float(p.grad.double().norm()) for pn, p in mod.named_parameters()}
        arrs[f'{name}.gradnorm_keys'] = np.array(sorted(gn))
        arrs[f'{name}.gradnorm'] = np.array([gn[q] for q in sorted(gn)])
        mod.eval()
        with torch.no_grad():
            arrs[f'{name}.eval_y'] = mod(x.clone())

    torch.manual_seed(7)
    run('resblock', ResBlock(32, num_blocks=2), recipe.randn((2, 32, 8, 8), 501), 601)
    run('csp0', CSPDownSample0(32, 64), recipe.randn((2, 32, 12, 12), 502), 602)
    run('csp', CSPDownSample(32, 64, num_blocks=2), recipe.randn((2, 32, 12, 12), 503), 603)
    # SPP has fixed 1024/512/2048 channels (yolov4.py:50-64): keep the map tiny
    run('spp', SPPBlock(), recipe.randn((1, 1024, 6, 6), 504), 604)
    up = Upsample()
    x = recipe.randn((2, 8, 3, 3), 505)
    up.train(); arrs['up.train'] = up(x, (2, 8, 6, 6))
    up.eval(); arrs['up.eval'] = up(x, (2, 8, 6, 6))
    arrs['up.x'] = x
    save('blocks.npz', **arrs)


# --------------------------------------------------------------------------- E
def gen_model():
    SEED = 1234
    torch.manual_seed(0)
    model = YOLOv4(recipe.MODEL_CFG, device=CPU)
    sd = model.state_dict()
    keys = list(sd.keys())
    shapes = [tuple(sd[k].shape) for k in keys]
    recipe.fill_state_dict_(sd, SEED)
    model.load_state_dict(sd)
    arrs = {'keys': np.array(keys), 'shapes': np.array([str(s) for s in shapes]), 'seed': np.array(SEED)}

    # ---- eval: calibrate (train fwd, momentum 1) then eval forward at S=64 and S=128
    x_cal = recipe.randn((8, 3, 64, 64), 77)
    recipe.calibrate_bn_(model, x_cal)
    model.eval()
    rm = model.state_dict()
    # pins train-mode BN statistics deep into the net
    for k in ('backbone.stem.norm.running_mean', 'backbone.stem.norm.running_var',
              'backbone.stage3.part2.1.module_list.7.1.norm.running_var',
              'backbone.stage5.transition.norm.running_mean', 'neck.spp.conv2.norm.running_var',
              'neck.pan.module2.4.norm.running_mean', 'head.yolo3.0.norm.running_var'):
        arrs['cal.' + k] = rm[k].clone()
    for S, B, seed in ((64, 2, 78), (128, 1, 79)):
        if S == 128:   # BN statistics are resolution dependent at this depth: re-calibrate
            sd = model.state_dict()
            recipe.fill_state_dict_(sd, SEED)
            model.load_state_dict(sd)
            recipe.calibrate_bn_(model, recipe.randn((4, 3, 128, 128), 76))
            model.eval()
        x = recipe.randn((B, 3, S, S), seed)
        with torch.no_grad():
            out = model(x)
        arrs[f'eval{S}.out'] = out.clone()
        print(f'  eval S={S}: out {tuple(out.shape)} absmax {float(out.abs().max()):.3f} '
              f'score range [{float((out[..., 4:5] * out[..., 5:]).min()):.4f}, '
              f'{float((out[..., 4:5] * out[..., 5:]).max()):.4f}]')
        if S == 64:
            det = postprocess(out.clone(), 80, conf_thre=0.12, nms_thre=0.4)
            for b in range(B):
                d = det[b] if det[b] is not None else torch.zeros((0, 7))
                arrs[f'eval64.det{b}'] = d
                print(f'  eval64 img {b}: {len(d)} detections')

    # ---- train step at S=128, B=2: fresh recipe weights (running stats reset)
    sd = model.state_dict()
    recipe.fill_state_dict_(sd, SEED)
    model.load_state_dict(sd)
    model.train()
    S, B = 128, 2
    x = recipe.randn((B, 3, S, S), 80)
    labels = recipe.synth_labels(B, S, 81, counts=[9, 21])
    crit = YOLOLoss(recipe.MODEL_CFG, ignore_thresh=0.7, device=CPU)
    outs = model(x)
    for l in range(3):
        arrs[f'train128.output{l}'] = outs[l]['output'].detach().contiguous().clone()
        arrs[f'train128.pred{l}'] = outs[l]['pred'].detach().contiguous().clone()
    loss = crit(outs, {'padded_labels': labels.clone()})
    loss.backward()
    arrs['train128.loss'] = loss.detach()
    print(f'  train S=128: loss {float(loss):.4f}')
    gn_keys, gn = [], []
    for pn, p in model.named_parameters():
        gn_keys.append(pn)
        gn.append(float(p.grad.double().norm()))
    arrs['train128.gradnorm_keys'] = np.array(gn_keys)
    arrs['train128.gradnorm'] = np.array(gn)
    named = dict(model.named_parameters())
    for k in ('backbone.stem.conv.weight', 'backbone.stem.norm.weight', 'backbone.stem.norm.bias',
              'backbone.stage1.part2_1_2.1.conv.weight', 'backbone.stage3.part2.1.module_list.3.1.norm.weight',
              'backbone.stage5.part2.1.module_list.0.0.norm.bias', 'neck.fpn.conv10.norm.weight',
              'neck.pan.conv1.norm.bias', 'head.yolo1.1.conv.bias', 'head.yolo2.1.conv.bias',
              'head.yolo3.1.conv.bias', 'head.yolo2.1.conv.weight'):
        arrs['train128.grad.' + k] = named[k].grad.clone()
    g = named['neck.spp.conv1.1.conv.weight'].grad
    arrs['train128.gradslice.neck.spp.conv1.1.conv.weight'] = g[:8, :16].clone()
    after = model.state_dict()
    for k in ('backbone.stem.norm.running_mean', 'backbone.stem.norm.running_var',
              'head.yolo3.0.norm.running_var', 'head.yolo3.0.norm.num_batches_tracked'):
        arrs['train128.after.' + k] = after[k].clone()
    save('model.npz', **arrs)


def gen_coco():
    """validate()'s detection -> COCO-record arithmetic (yolo/engine/build.py:144-164): the reference's own
    yolobox2xywh on Python floats taken from fp32 detections.  (engine/build.py itself needs apex/pycocotools
    and cannot be imported; the per-detection calls are made here exactly as that loop makes them.)"""
    from yolo.util.utils import yolobox2xywh
    rng = np.random.RandomState(77)
    arrs = {}
    for ci, (sh, sw, S, n) in enumerate([(480, 640, 416, 23), (333, 500, 608, 40), (1216, 1216, 608, 5)]):
        x1 = rng.uniform(0, S * 0.8, n); y1 = rng.uniform(0, S * 0.8, n)
        det = np.stack([x1, y1, x1 + rng.uniform(2, S * 0.2, n), y1 + rng.uniform(2, S * 0.2, n),
                        rng.uniform(0.01, 1, n), rng.uniform(0.01, 1, n), rng.randint(0, 80, n)], 1).astype(np.float32)
        out = torch.from_numpy(det)
        info = [sh, sw, S, S]
        bbox, score = [], []
        for o in out:
            bbox.append(yolobox2xywh((float(o[1]), float(o[0]), float(o[3]), float(o[2])), info[:4]))
            score.append(float(o[4].data.item() * o[5].data.item()))
        arrs[f'c{ci}.det'] = det
        arrs[f'c{ci}.info'] = np.asarray(info)
        arrs[f'c{ci}.bbox'] = np.asarray(bbox, dtype=np.float64)
        arrs[f'c{ci}.score'] = np.asarray(score, dtype=np.float64)
    # the class-index -> COCO category-id table is data held by yolo/data/cocodataset.py (not importable: cv2);
    # read the literal out of the file's text
    import ast
    import re
    txt = open('/root/reference/yolo/data/cocodataset.py').read()
    arrs['class_ids'] = np.asarray(ast.literal_eval(re.search(r'coco_class_ids = (\[[^\]]*\])', txt).group(1)))
    save('coco.npz', **arrs)


GENS = {'coco': gen_coco, 'iou_nms': gen_iou_nms, 'yololayer': gen_yololayer, 'yololoss': gen_yololoss,
        'postprocess': gen_postprocess, 'convbnact': gen_convbnact, 'blocks': gen_blocks, 'model': gen_model}

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('which', nargs='*', default=list(GENS))
    for w in ap.parse_args().which:
        print(f'== {w}')
        GENS[w]()
