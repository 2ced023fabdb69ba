import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipe, yolov4_amd
from yolov4_amd import ops
yolov4_amd.set_conv_mode('bf16')
dev = torch.device('cuda:0')
cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
bf = lambda t: t.bfloat16().double()
for (B, ci, co, k, H) in [(2, 64, 128, 3, 9), (2, 128, 128, 1, 12), (2, 64, 64, 1, 8), (2, 256, 256, 1, 8)]:
    x = recipe.randn((B, ci, H, H), 7); dy = recipe.randn((B, co, H, H), 9)
    xp = ops.planes_split_raw(cl(x)); dyp = ops.planes_split_raw(cl(dy))
    dw = ops.conv_wgrad_planes_raw(xp, dyp, (co, ci, k, k), k).double().cpu()
    print(ops.last_conv_kernel())
    ref = torch.nn.grad.conv2d_weight(bf(x), (co, ci, k, k), bf(dy), 1, (k - 1) // 2)
    err = (dw - ref).abs()
    print((B, ci, co, k, H), 'max err', float(err.max()), 'ref max', float(ref.abs().max()))
    e = err / ref.abs().max()
    bad = e > 1e-5
    print(' bad frac', float(bad.float().mean()))
    print(' bad by n%16', [float(bad[i::16].float().mean()) for i in range(16)])
    print(' bad by n//16', [round(float(bad[i*16:(i+1)*16].float().mean()), 2) for i in range(co // 16)])
    print(' bad by c//16', [round(float(bad[:, i*16:(i+1)*16].float().mean()), 2) for i in range(ci // 16)])
    print(' bad by c%16', [round(float(bad[:, i::16].float().mean()), 2) for i in range(16)])
    if k == 3:
        print(' bad by tap', [round(float(bad[:, :, r, q].float().mean()), 2) for r in range(3) for q in range(3)])
    # is dw a permutation? compare with ref shifted in c
    if float(bad.float().mean()) > 0:
        n0 = 0
        row, rrow = dw[n0, :, 0, 0], ref[n0, :, 0, 0]
        print(' dw[0,:8]', row[:8].tolist()); print(' ref[0,:8]', rrow[:8].tolist())
        # find for dw[0,c] best matching ref[0,c']
        m = [(int((rrow - row[c]).abs().argmin()), float((rrow - row[c]).abs().min())) for c in range(min(ci, 32))]
        print(' match', m)
