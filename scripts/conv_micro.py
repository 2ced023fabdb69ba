"""Micro-benchmark of the conv kernels on BASELINE-size layers (B=64 @608 geometry).
usage: python scripts/conv_micro.py [reps] [kinds]      kinds subset of fwd,dgrad,wgrad"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import yolov4_amd
from yolov4_amd import ops
if os.environ.get('CONV_MODE'):
    yolov4_amd.set_conv_mode(os.environ['CONV_MODE'])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
kinds = sys.argv[2].split(',') if len(sys.argv) > 2 else ['fwd', 'dgrad', 'wgrad']
only = sys.argv[3] if len(sys.argv) > 3 else None
dev = torch.device('cuda:0')
B = 64
SHAPES = [  # Cin, Cout, k, s, H
    (128, 128, 3, 1, 76), (256, 256, 3, 1, 38), (512, 512, 3, 1, 19), (256, 512, 3, 1, 38), (512, 1024, 3, 1, 19),
    (128, 256, 3, 1, 76), (64, 64, 1, 1, 304), (512, 256, 1, 1, 38), (64, 128, 3, 2, 304), (32, 64, 3, 2, 608),
    (256, 512, 3, 2, 76), (64, 64, 3, 1, 152), (128, 128, 1, 1, 76),
    (64, 64, 1, 1, 152), (32, 64, 3, 1, 304), (64, 32, 1, 1, 304),
    (256, 128, 1, 1, 76), (1024, 512, 1, 1, 19), (256, 256, 1, 1, 38), (128, 64, 1, 1, 304), (512, 512, 1, 1, 19),
    (3, 32, 3, 1, 608),
]
if only:
    idx = [int(i) for i in only.split(',')]
    SHAPES = [SHAPES[i] for i in idx]
g = torch.Generator(device='cpu'); g.manual_seed(0)
for (ci, co, k, s, H) in SHAPES:
    x = torch.randn((B, ci, H, H), generator=g).to(dev)
    if ci != 3:
        x = x.contiguous(memory_format=torch.channels_last)
    w = (torch.randn((co, ci, k, k), generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
    dy = torch.randn((B, co, Ho, Ho), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * Ho * Ho * co * ci * k * k
    res = []
    for kind in kinds:
        if ci == 3 and kind == 'dgrad':
            continue
        fn = {'fwd': lambda: ops.conv_fwd_raw(x, w, k, s), 'dgrad': lambda: ops.conv_dgrad_raw(dy, w, (B, ci, H, H), k, s),
              'wgrad': lambda: ops.conv_wgrad_raw(x, dy, (co, ci, k, k), k, s)}[kind]
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append(f'{kind} {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF')
    print(f'{ci:5d}->{co:5d} k{k} s{s} @{H:3d}: ' + ' | '.join(res), flush=True)
    del x, w, dy
