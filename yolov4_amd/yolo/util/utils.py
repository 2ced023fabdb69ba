# -*- coding: utf-8 -*-
"""Post-processing behind the reference's API (yolo/util/utils.py:32-89, 92-223):
confidence filter + per-class greedy NMS on the GPU.

One host round trip per call (per-image offsets of the compacted result) replaces the
reference's per-class device->host copies and its Python/numpy loops: candidate prefix
sums, key fill, sort + NMS and compaction all run on the device.  Tie order is
DEFINED here (score desc, then lower box index first); the reference's comes from an
unstable numpy argsort and is platform dependent.
"""
import ctypes

import numpy as np
import torch

from ... import ops
from ..._lib import check, lib


def _dev():
    if not torch.cuda.is_available():
        raise ops.Y4Error('postprocess / nms run on an MI355X only (no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


def nms(bbox, thresh, score=None, limit=None):
    """bbox [R,4] xyxy, numpy in / numpy int32 indices out, as utils.py:32-89."""
    bbox = np.asarray(bbox, dtype=np.float32)
    if len(bbox) == 0:
        return np.zeros((0,), dtype=np.int32)
    L = lib()
    dev = _dev()
    R = bbox.shape[0]
    boxes = torch.from_numpy(np.ascontiguousarray(bbox)).to(dev)
    sc = torch.from_numpy(np.ascontiguousarray(np.asarray(score, dtype=np.float32))).to(dev) if score is not None else None
    keep = torch.empty(R, dtype=torch.int32, device=dev)
    nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
    nbytes = L.y4_nms_workspace(R)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(L.y4_nms_f32(ops._ptr(boxes), ops._ptr(sc), R, float(thresh), int(limit) if limit is not None else 0,
                       ops._ptr(keep), ops._ptr(nkeep), ops._ptr(ws), nbytes, ops._stream()), 'nms')
    n = int(nkeep.item())
    return keep[:n].cpu().numpy().astype(np.int32)


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45):
    """prediction [B,N,5+C] (xc,yc,w,h,obj,cls...) -> list of [n,7] tensors
    (x1,y1,x2,y2,obj,cls_conf,cls_id) or None; rows ordered class asc, score desc.
    Side effect kept from the reference: prediction[:, :, :4] becomes xyxy in place."""
    L = lib()
    src = prediction
    on_host = not prediction.is_cuda
    if on_host:                       # detect.py:115-118 moves the output to the CPU first
        prediction = prediction.to(_dev())
    if prediction.dtype != torch.float32 or not prediction.is_contiguous():
        raise ops.Y4Error('postprocess expects a contiguous float32 [B,N,5+C] tensor')
    B, N, n_ch = prediction.shape
    if n_ch != 5 + num_classes:
        raise ops.Y4Error('postprocess: last dim must be 5 + num_classes')
    dev = prediction.device
    out = [None for _ in range(B)]
    if N == 0:
        return out
    nseg = B * num_classes
    counts = torch.empty(nseg, dtype=torch.int32, device=dev)
    st = ops._stream()
    check(L.y4_post_count_f32(ops._ptr(prediction), B, N, num_classes, float(conf_thre), 1, ops._ptr(counts), st),
          'post_count')
    if on_host:
        src[:, :, :4] = prediction[:, :, :4].cpu()
    # Everything between the count and the result stays on the device: prefix sums, key fill, sort + NMS, compaction run
    # into buffers sized for `cap` candidates; the ONE host read at the end brings the per-image offsets and the true
    # candidate total (if it exceeded cap -- many classes per box above a low threshold -- the sequence repeats with room).
    cap = int(min(B * N * num_classes, max(B * N // 2, 1 << 16), (1 << 31) - 1))
    seg_off = torch.empty(nseg + 1, dtype=torch.int32, device=dev)
    meta = torch.empty(2 + B + 1, dtype=torch.int32, device=dev)          # [total, overflow | img_off[0..B]]
    out_off = torch.empty(nseg + 1, dtype=torch.int32, device=dev)
    kept = torch.empty(nseg, dtype=torch.int32, device=dev)
    while True:
        check(L.y4_post_scan_i32(ops._ptr(counts), nseg, cap, ops._ptr(seg_off), ops._ptr(meta), st), 'post_scan')
        rows = torch.empty((cap, 7), dtype=torch.float32, device=dev)
        final = torch.empty((cap, 7), dtype=torch.float32, device=dev)
        nbytes = L.y4_post_nms_workspace(cap, nseg)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        check(L.y4_post_nms_f32(ops._ptr(prediction), B, N, num_classes, float(conf_thre), float(nms_thre),
                                ops._ptr(seg_off), cap, ops._ptr(rows), ops._ptr(kept), ops._ptr(ws), nbytes, st), 'post_nms')
        check(L.y4_post_compact_f32(ops._ptr(rows), ops._ptr(seg_off), ops._ptr(kept), B, num_classes, ops._ptr(final),
                                    ops._ptr(out_off), ops._ptr(meta[2:]), st), 'post_compact')
        host = meta.cpu().numpy().astype(np.int64)                        # the host sync
        if not host[1]:
            break
        cap = int(host[0])
    img_off = host[2:]
    n_det = int(img_off[B])
    if n_det == 0:
        return out
    dets = final[:n_det].clone()                                          # (lets the cap-sized buffers go)
    if on_host:
        dets = dets.cpu()
    for b in range(B):
        if img_off[b + 1] > img_off[b]:
            out[b] = dets[img_off[b]:img_off[b + 1]]
    return out


# ------------------------------------------------------------------ detections -> COCO records
# the 80 COCO category ids in class-index order (yolo/data/cocodataset.py:48-51): 1..90 minus the unused ids
COCO_CLASS_IDS = [i for i in range(1, 91) if i not in (12, 26, 29, 30, 45, 66, 68, 69, 71, 83)]


def yolobox2xywh(box, info_img):
    """(y1, x1, y2, x2) in network-input pixels -> [x1, y1, w, h] in source-image pixels
    (yolo/util/utils.py:281-309); `info_img` = (src_h, src_w, dst_h, dst_w)."""
    src_h, src_w, dst_h, dst_w = info_img
    y1, x1, y2, x2 = box
    box_h = (y2 - y1) / dst_h * src_h
    box_w = (x2 - x1) / dst_w * src_w
    return [x1 / dst_w * src_w, y1 / dst_h * src_h, box_w, box_h]


def detections_to_coco(detections, img_info, image_id, class_ids=COCO_CLASS_IDS):
    """One image's postprocess() result [n,7] (or None) -> list of COCO result dicts, as validate() builds
    them (yolo/engine/build.py:144-164).  The reference converts every fp32 number to a Python float and
    does the arithmetic in double precision; one device->host copy and the same double arithmetic here."""
    if detections is None:
        return []
    d = detections.detach().cpu().numpy().astype(np.float64)
    out = []
    for r in d:
        out.append({'image_id': image_id,
                    'category_id': class_ids[int(r[6])],
                    'bbox': yolobox2xywh((float(r[1]), float(r[0]), float(r[3]), float(r[2])), img_info[:4]),
                    'score': float(r[4] * r[5]),
                    'segmentation': []})
    return out
