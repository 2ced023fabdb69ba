# -*- coding: utf-8 -*-
"""Times the eval decode of the three YOLO layers at bs = 32 @608 (HIP events, in-process)."""
import sys
import torch
sys.path.insert(0, '.')
from yolov4_amd import ops
dev = torch.device('cuda:0')
B = 32
for F, stride in ((76, 8), (38, 16), (19, 32)):
    lg = torch.randn(B, 256, F, F, device=dev).contiguous(memory_format=torch.channels_last)[:, :255]
    anchors = [(12, 16), (19, 36), (40, 28)]
    out = torch.empty((B, 3 * F * F, 85), device=dev)
    for _ in range(3):
        ops.yolo_decode_eval(lg, anchors, 80, stride, out=out, n_total=3 * F * F, box_off=0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.yolo_decode_eval(lg, anchors, 80, stride, out=out, n_total=3 * F * F, box_off=0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    byt = 2 * B * 3 * F * F * 85 * 4
    print(f'F={F}: {ms * 1e3:.1f} us, {byt / ms / 1e9:.2f} TB/s (algorithmic)')
