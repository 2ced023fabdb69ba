# -*- coding: utf-8 -*-
"""A/B on one box: DMA-fed plane kernels (conv_planes.hip) vs the register-staged f16x2 kernels on the layer shapes of the
bs = 64 @608 step.  Interleaved rounds in one process (guide rule 24); both arms include the per-call filter split."""
import json
import sys

import numpy as np
import torch

import os  # noqa: E402
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from yolov4_amd import ops  # noqa: E402

SHAPES = [  # ci, co, k, s, H
    (128, 128, 3, 1, 76), (256, 256, 3, 1, 38), (512, 512, 3, 1, 19), (128, 256, 3, 1, 76), (256, 512, 3, 1, 38),
    (512, 1024, 3, 1, 19), (64, 64, 3, 1, 152), (256, 128, 1, 1, 76), (512, 256, 1, 1, 38), (1024, 512, 1, 1, 19),
    (128, 128, 1, 1, 76), (256, 256, 1, 1, 38), (128, 256, 3, 2, 152), (256, 512, 3, 2, 76),
]


# the stage-1 / stage-2 layers with 64 output channels (half-empty 128-column tiles): Y4_MICRO_SHAPES=small
SMALL = [(64, 64, 1, 1, 304), (64, 64, 1, 1, 152), (128, 64, 1, 1, 152), (64, 64, 3, 1, 152), (128, 64, 1, 1, 304)]

MODE = 'fwd'


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    global MODE
    MODE = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
    only = [int(i) for i in sys.argv[3].split(',')] if len(sys.argv) > 3 else None      # indices into SHAPES
    dev = torch.device('cuda:0')
    out = []
    shapes = SMALL if os.environ.get('Y4_MICRO_SHAPES') == 'small' else SHAPES
    ybf = False
    if os.environ.get('Y4_MICRO_BF16') == '1':       # plane arm: bf16 operands (and results); register-staged arm: f16x2 (the hybrid)
        import yolov4_amd
        yolov4_amd.set_conv_mode('bf16')
        ybf = True
    for ci, co, k, s, H in ([shapes[i] for i in only] if only else shapes):
        x = torch.randn(B, ci, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, k, k, device=dev) / np.sqrt(ci * k * k)).contiguous(memory_format=torch.channels_last)
        xa = ops.amax_raw(x)
        xp = ops.planes_split_raw(x, xa)
        Ho0 = (H + 2 * ((k - 1) // 2) - k) // s + 1
        if MODE in ('fwd_agpr', 'dgrad_agpr'):
            # experiment (experiments/agpr_conv/README.md; needs that kernel compiled into the library): 'regstage' column = the
            # product plane kernel, 'planes' column = the AGPR-accumulator variant
            def with_env(val, f):
                def run():
                    os.environ['Y4_PLANES_AGPR'] = val
                    try:
                        return f()
                    finally:
                        os.environ['Y4_PLANES_AGPR'] = '0'
                return run
            if MODE == 'fwd_agpr':
                f = lambda: ops.conv_fwd_planes_raw(xp, w, k, s)
            else:
                if s != 1:
                    continue
                dy = torch.randn(B, co, Ho0, Ho0, device=dev).contiguous(memory_format=torch.channels_last)
                dyp = ops.planes_split_raw(dy, ops.amax_raw(dy))
                f = lambda: ops.conv_dgrad_planes_raw(dyp, w, (B, ci, H, H), k)
            arms = {'regstage': with_env('0', f), 'planes': with_env('1', f)}
            ya = arms['regstage'](); yb = arms['planes']()
            ya = ya[0] if isinstance(ya, tuple) else ya; yb = yb[0] if isinstance(yb, tuple) else yb
            torch.cuda.synchronize()
            print('max |agpr - product| / max:', float((ya - yb).abs().max() / ya.abs().max()), ops.last_conv_kernel(), flush=True)
        elif MODE == 'fwd':
            arms = {'regstage': lambda: ops.conv_fwd_bnstats_raw(x, w, k, s, None, None, None, 0.1, 1e-5, x_amax=xa),
                    'planes': lambda: ops.conv_fwd_planes_raw(xp, w, k, s, y_bf16=ybf and co % 32 == 0)}
        else:
            if s != 1:
                continue
            dy = torch.randn(B, co, Ho0, Ho0, device=dev).contiguous(memory_format=torch.channels_last)
            da = ops.amax_raw(dy)
            dyp = ops.planes_split_raw(dy, da)
            if MODE == 'dgrad':
                arms = {'regstage': lambda: ops.conv_dgrad_raw(dy, w, (B, ci, H, H), k, s, dy_amax=da),
                        'planes': lambda: ops.conv_dgrad_planes_raw(dyp, w, (B, ci, H, H), k)}
            elif MODE == 'dgrad_res':
                # skip-operand epilogue: 'regstage' column = the plane kernel WITHOUT a residual, 'planes' = with one
                res = torch.randn(B, ci, H, H, device=dev).contiguous(memory_format=torch.channels_last)
                arms = {'regstage': lambda: ops.conv_dgrad_planes_raw(dyp, w, (B, ci, H, H), k),
                        'planes': lambda: ops.conv_dgrad_planes_raw(dyp, w, (B, ci, H, H), k, residual=res)}
            else:
                arms = {'regstage': lambda: ops.conv_wgrad_raw(x, dy, (co, ci, k, k), k, s, x_amax=xa, dy_amax=da),
                        'planes': lambda: ops.conv_wgrad_planes_raw(xp, dyp, (co, ci, k, k), k)}
        for f in arms.values():
            f()
        torch.cuda.synchronize()
        ts = {a: [] for a in arms}
        for _ in range(6):
            for a, f in arms.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    f()
                e1.record()
                torch.cuda.synchronize()
                ts[a].append(e0.elapsed_time(e1) / 5)
        Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
        fl = 2.0 * B * Ho * Ho * co * ci * k * k
        row = {'shape': f'{ci}->{co} k{k} s{s} @{H}', 'gflop': fl / 1e9}
        for a in arms:
            med = float(np.median(ts[a]))
            row[a + '_ms'] = round(med, 4)
            row[a + '_tflops'] = round(fl / med / 1e9, 1)
        row['speedup'] = round(row['regstage_ms'] / row['planes_ms'], 3)
        print(json.dumps(row), flush=True)
        out.append(row)
        del x, w, xp


if __name__ == '__main__':
    main()
