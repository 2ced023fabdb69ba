# -*- coding: utf-8 -*-
"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- numpy restatement of the YOLO
head decode, the detection loss and the post-processing of the reference.

All arithmetic is float32 (np.float32) unless stated, mirroring the reference's
fp32 torch CPU path; reductions of the loss are carried in float64 and compared
with a relative tolerance (the reference sums ~1e5-magnitude fp32 values,
SURVEY.md D9).
"""
import numpy as np

F32 = np.float32
STRIDES = (8, 16, 32)          # yolo/model/yololayer.py:54, yolo/model/yololoss.py:99


def _sigmoid(x):
    x = x.astype(F32)
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def masked_anchors(cfg, layer_no):
    """yolo/model/yololayer.py:65-76: anchors / stride in float64, then cast to
    the activation dtype (fp32) at use (:117-120)."""
    st = STRIDES[layer_no]
    grid = [(w / st, h / st) for w, h in cfg['ANCHORS']]
    return np.array([grid[i] for i in cfg['ANCHOR_MASK'][layer_no]], dtype=np.float64).astype(F32)


def all_anchors(cfg, layer_no):
    st = STRIDES[layer_no]
    return np.array([(w / st, h / st) for w, h in cfg['ANCHORS']], dtype=np.float64).astype(F32)


# ------------------------------------------------------------------ decode
def yolo_decode(logits, layer_no, cfg, train):
    """yolo/model/yololayer.py:88-166.  logits [B, 3*(5+C), F, F] fp32 (NCHW).
    train -> (output [B,3,F,F,5+C], pred [B,3,F,F,4]) in grid units;
    eval  -> [B, 3*F*F, 5+C] in input pixels, box index a*F*F + j*F + i."""
    logits = np.asarray(logits, dtype=F32)
    B, ch, Fs, _ = logits.shape
    n_ch = 5 + cfg['N_CLASSES']
    A = ch // n_ch
    out = logits.reshape(B, A, n_ch, Fs, Fs).transpose(0, 1, 3, 4, 2).copy()     # :101
    sig = np.r_[0:2, 4:n_ch]
    out[..., sig] = _sigmoid(out[..., sig])                                     # :105
    xs = np.arange(Fs, dtype=F32).reshape(1, 1, 1, Fs)                          # :109-113
    ys = np.arange(Fs, dtype=F32).reshape(1, 1, Fs, 1)
    anc = masked_anchors(cfg, layer_no)
    aw = anc[:, 0].reshape(1, A, 1, 1)
    ah = anc[:, 1].reshape(1, A, 1, 1)
    pred = out.copy()
    pred[..., 0] += xs
    pred[..., 1] += ys
    pred[..., 2] = np.exp(pred[..., 2], dtype=F32) * aw
    pred[..., 3] = np.exp(pred[..., 3], dtype=F32) * ah
    if train:
        return out, pred[..., :4].copy()                                        # :136-145
    pred[..., :4] *= F32(STRIDES[layer_no])                                     # :162
    return pred.reshape(B, -1, n_ch)                                            # :166


def yolo_decode_backward(logits, g_out, g_pred, layer_no, cfg):
    """d(logits) for train mode given d(output), d(pred) (either may be None)."""
    logits = np.asarray(logits, dtype=F32)
    B, ch, Fs, _ = logits.shape
    n_ch = 5 + cfg['N_CLASSES']
    A = ch // n_ch
    out, pred = yolo_decode(logits, layer_no, cfg, True)
    g = np.zeros_like(out)
    if g_out is not None:
        g += np.asarray(g_out, dtype=F32)
    if g_pred is not None:
        gp = np.asarray(g_pred, dtype=F32)
        g[..., 0] += gp[..., 0]
        g[..., 1] += gp[..., 1]
        g[..., 2] += gp[..., 2] * pred[..., 2]      # d/dt (exp(t)*a) = pred
        g[..., 3] += gp[..., 3] * pred[..., 3]
    sig = np.r_[0:2, 4:n_ch]
    o = out[..., sig]
    g[..., sig] = g[..., sig] * ((F32(1) - o) * o)  # torch sigmoid_backward: grad * (1-y) * y
    return g.transpose(0, 1, 4, 2, 3).reshape(B, ch, Fs, Fs).copy()


# ------------------------------------------------------------------ IoU
def bboxes_iou(a, b, xyxy=True):
    """yolo/model/yololoss.py:16-91 (pairwise [Na,4] x [Nb,4] -> [Na,Nb])."""
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    if a.shape[1] != 4 or b.shape[1] != 4:
        raise IndexError
    if xyxy:
        tl = np.maximum(a[:, None, :2], b[None, :, :2])
        br = np.minimum(a[:, None, 2:], b[None, :, 2:])
        area_a = np.prod(a[:, 2:] - a[:, :2], 1, dtype=F32)
        area_b = np.prod(b[:, 2:] - b[:, :2], 1, dtype=F32)
    else:
        tl = np.maximum(a[:, None, :2] - a[:, None, 2:] / F32(2), b[None, :, :2] - b[None, :, 2:] / F32(2))
        br = np.minimum(a[:, None, :2] + a[:, None, 2:] / F32(2), b[None, :, :2] + b[None, :, 2:] / F32(2))
        area_a = np.prod(a[:, 2:], 1, dtype=F32)
        area_b = np.prod(b[:, 2:], 1, dtype=F32)
    en = (tl < br).astype(F32).prod(axis=2)
    area_i = np.prod(br - tl, 2, dtype=F32) * en
    with np.errstate(divide='ignore', invalid='ignore'):
        return (area_i / (area_a[:, None] + area_b[None, :] - area_i)).astype(F32)


# ------------------------------------------------------------------ targets
def build_target(output, pred, layer_no, labels, cfg, ignore_thresh=0.7):
    """yolo/model/yololoss.py:118-371.  Returns (target [B,A,F,F,5+C],
    obj_mask [B,A,F,F], tgt_mask [B,A,F,F,4+C], tgt_scale [B,A,F,F,2]).
    Sequential semantics preserved: ignore mask first, then positives in label
    order (last writer wins for xy/wh/scale, class bits accumulate)."""
    output = np.asarray(output, dtype=F32)
    pred = np.asarray(pred, dtype=F32)
    B, A, Fs = output.shape[0], output.shape[1], output.shape[2]
    C = cfg['N_CLASSES']
    n_ch = 5 + C
    labels = np.asarray(labels).astype(F32)                                     # :129 labels.to(dtype)
    st = F32(STRIDES[layer_no])
    anch_mask = cfg['ANCHOR_MASK'][layer_no]
    ref = all_anchors(cfg, layer_no)                                            # [9,2] fp32
    manc = masked_anchors(cfg, layer_no)                                        # torch.Tensor(list) -> fp32

    tgt_mask = np.zeros((B, A, Fs, Fs, 4 + C), F32)
    obj_mask = np.ones((B, A, Fs, Fs), F32)
    tgt_scale = np.zeros((B, A, Fs, Fs, 2), F32)
    target = np.zeros((B, A, Fs, Fs, n_ch), F32)

    tx_all = labels[:, :, 0] / st
    ty_all = labels[:, :, 1] / st
    tw_all = labels[:, :, 2] / st
    th_all = labels[:, :, 3] / st
    ti_all = tx_all.astype(np.int16)
    tj_all = ty_all.astype(np.int16)
    nlabel = (labels.sum(axis=2, dtype=F32) > 0).sum(axis=1)                    # :219

    ref_box = np.zeros((ref.shape[0], 4), F32)
    ref_box[:, 2:] = ref
    for b in range(B):
        n = int(nlabel[b])
        if n == 0:
            continue
        tb = np.zeros((n, 4), F32)
        tb[:, 2] = tw_all[b, :n]
        tb[:, 3] = th_all[b, :n]
        aiou = bboxes_iou(tb, ref_box, xyxy=True)                               # :249
        best_all = np.argmax(aiou, axis=1)                                      # first max
        best_n = best_all % 3
        best_mask = (best_all == anch_mask[0]) | (best_all == anch_mask[1]) | (best_all == anch_mask[2])
        tb[:, 0] = tx_all[b, :n]
        tb[:, 1] = ty_all[b, :n]
        piou = bboxes_iou(pred[b].reshape(-1, 4), tb, xyxy=False)               # :276
        # torch.max over dim propagates NaN; NaN > thresh is False
        has_nan = np.isnan(piou).any(axis=1)
        with np.errstate(invalid='ignore'):
            best = np.where(has_nan, F32(np.nan), np.nanmax(np.where(np.isnan(piou), -np.inf, piou), axis=1))
            ign = best > F32(ignore_thresh)
        obj_mask[b] = (~ign).reshape(A, Fs, Fs).astype(F32)                     # :294
        if best_mask.sum() == 0:
            continue
        for t in range(n):
            if not best_mask[t]:
                continue
            i, j, a = int(ti_all[b, t]), int(tj_all[b, t]), int(best_n[t])
            obj_mask[b, a, j, i] = 1
            tgt_mask[b, a, j, i, :] = 1
            tgt_scale[b, a, j, i, :] = np.sqrt(F32(2) - tw_all[b, t] * th_all[b, t] / F32(Fs) / F32(Fs))
            target[b, a, j, i, 0] = tx_all[b, t] - F32(ti_all[b, t])
            target[b, a, j, i, 1] = ty_all[b, t] - F32(tj_all[b, t])
            target[b, a, j, i, 2] = np.log(tw_all[b, t] / manc[a, 0] + F32(1e-16))
            target[b, a, j, i, 3] = np.log(th_all[b, t] / manc[a, 1] + F32(1e-16))
            target[b, a, j, i, 4] = 1
            target[b, a, j, i, 5 + int(labels[b, t, 4].astype(np.int16))] = 1
    return target, obj_mask, tgt_mask, tgt_scale


def _bce_terms(o, t):
    """nn.BCELoss elementwise: -(t*max(log o,-100) + (1-t)*max(log(1-o),-100))."""
    with np.errstate(divide='ignore'):
        lo = np.maximum(np.log(o.astype(F32)), F32(-100))
        l1 = np.maximum(np.log((F32(1) - o).astype(F32)), F32(-100))
    return -(t * lo + (F32(1) - t) * l1)


def _bce_grad(o, t):
    """torch binary_cross_entropy_backward: (o - t) / max((1-o)*o, 1e-12)."""
    return (o - t) / np.maximum((F32(1) - o) * o, F32(1e-12))


def yolo_loss_layer(output, pred, layer_no, labels, cfg, ignore_thresh=0.7):
    """yolo/model/yololoss.py:385-432 for one layer.  Returns dict with the
    scalar loss (float64), its 4 parts, grad wrt `output` (fp32, the un-masked
    sigmoid/raw tensor the YOLOLayer produced), the masks and the output tensor
    as the reference leaves it after its in-place masking side effect."""
    output = np.asarray(output, dtype=F32).copy()
    target, obj_mask, tgt_mask, tgt_scale = build_target(output, pred, layer_no, labels, cfg, ignore_thresh)
    n_ch = output.shape[-1]
    rest = np.r_[0:4, 5:n_ch]
    mo = output.copy()
    mo[..., 4] *= obj_mask                                                      # :402
    mo[..., rest] *= tgt_mask                                                   # :405
    mo[..., 2:4] *= tgt_scale                                                   # :407
    mt = target.copy()
    mt[..., 4] *= obj_mask
    mt[..., rest] *= tgt_mask
    mt[..., 2:4] *= tgt_scale
    w = tgt_scale * tgt_scale
    l_xy = (w * _bce_terms(mo[..., :2], mt[..., :2])).sum(dtype=np.float64)     # :417-421
    l_wh = ((mo[..., 2:4] - mt[..., 2:4]) ** 2).sum(dtype=np.float64) / 2       # :423
    l_obj = _bce_terms(mo[..., 4], mt[..., 4]).sum(dtype=np.float64)            # :425
    l_cls = _bce_terms(mo[..., 5:], mt[..., 5:]).sum(dtype=np.float64)          # :427
    # gradient wrt the un-masked output (chain through the in-place mask multiplies)
    g = np.zeros_like(output)
    g[..., :2] = w * _bce_grad(mo[..., :2], mt[..., :2]) * tgt_mask[..., :2]
    g[..., 2:4] = (mo[..., 2:4] - mt[..., 2:4]) * tgt_scale * tgt_mask[..., 2:4]
    g[..., 4] = _bce_grad(mo[..., 4], mt[..., 4]) * obj_mask
    g[..., 5:] = _bce_grad(mo[..., 5:], mt[..., 5:]) * tgt_mask[..., 4:]
    return dict(loss=l_xy + l_wh + l_obj + l_cls, xy=l_xy, wh=l_wh, obj=l_obj, cls=l_cls,
                grad_output=g, target=target, obj_mask=obj_mask, tgt_mask=tgt_mask, tgt_scale=tgt_scale,
                mutated_output=mo)


def yolo_loss(logits_list, labels, cfg, ignore_thresh=0.7):
    """Whole criterion on head logits [B,255,F,F] x3: decode (train) -> per-layer
    loss -> sum (yololoss.py:443).  Returns (loss float64, [d loss / d logits])."""
    total = 0.0
    grads = []
    for l, lg in enumerate(logits_list):
        out, pred = yolo_decode(lg, l, cfg, True)
        r = yolo_loss_layer(out, pred, l, labels, cfg, ignore_thresh)
        total += r['loss']
        grads.append(yolo_decode_backward(lg, r['grad_output'], None, l, cfg))
    return total, grads


# ------------------------------------------------------------------ NMS
def nms(bbox, thresh, score=None, limit=None):
    """yolo/util/utils.py:32-89 with a DEFINED tie order (score desc, ties:
    lower index first); fp32 arithmetic, suppression on IoU >= thresh, NaN
    IoU keeps the box."""
    bbox = np.asarray(bbox, dtype=F32)
    if len(bbox) == 0:
        return np.zeros((0,), dtype=np.int32)
    if score is not None:
        order = np.argsort(-np.asarray(score, dtype=F32), kind='stable')
        bbox = bbox[order]
    area = ((bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1])).astype(F32)
    keep = []
    kb = np.zeros((0, 4), F32)
    ka = np.zeros((0,), F32)
    thresh = F32(thresh)
    for i in range(len(bbox)):
        b = bbox[i]
        if len(keep):
            tl = np.maximum(b[:2], kb[:, :2])
            br = np.minimum(b[2:], kb[:, 2:])
            inter = ((br[:, 0] - tl[:, 0]) * (br[:, 1] - tl[:, 1])).astype(F32) * (tl < br).all(axis=1)
            with np.errstate(divide='ignore', invalid='ignore'):
                iou = inter / (area[i] + ka - inter)
                if (iou >= thresh).any():
                    continue
        keep.append(i)
        kb = np.concatenate([kb, b[None]], 0)
        ka = np.concatenate([ka, area[i:i + 1]], 0)
        if limit is not None and len(keep) >= limit:
            break
    keep = np.asarray(keep, dtype=np.int64)
    if score is not None:
        keep = order[keep]
    return keep.astype(np.int32)


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45):
    """yolo/util/utils.py:92-223.  prediction [B,N,5+C] fp32 numpy, MUTATED in
    place (xywh -> xyxy, :117-126).  Returns list of [n,7] fp32 arrays
    (x1,y1,x2,y2,obj,cls_conf,cls_id) or None; rows ordered class asc, score desc."""
    p = prediction
    assert p.dtype == F32
    corner = np.empty_like(p[:, :, :4])
    corner[:, :, 0] = p[:, :, 0] - p[:, :, 2] / F32(2)
    corner[:, :, 1] = p[:, :, 1] - p[:, :, 3] / F32(2)
    corner[:, :, 2] = p[:, :, 0] + p[:, :, 2] / F32(2)
    corner[:, :, 3] = p[:, :, 1] + p[:, :, 3] / F32(2)
    p[:, :, :4] = corner
    out = [None] * len(p)
    conf = F32(conf_thre)
    for b in range(len(p)):
        ip = p[b]
        sc = ip[:, 5:5 + num_classes] * ip[:, 4:5]
        box_idx, cls_idx = np.nonzero(sc >= conf)                               # row-major (box, class), :170
        if len(box_idx) == 0:
            continue
        det = np.concatenate([ip[box_idx, :5], ip[box_idx, 5 + cls_idx][:, None],
                              cls_idx.astype(F32)[:, None]], 1).astype(F32)
        rows = []
        for c in np.unique(cls_idx):
            dc = det[det[:, 6] == c]
            keep = nms(dc[:, :4], nms_thre, score=dc[:, 4] * dc[:, 5])
            rows.append(dc[keep])
        out[b] = np.concatenate(rows, 0)
    return out
