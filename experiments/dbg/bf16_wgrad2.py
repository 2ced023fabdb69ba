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
B, ci, co, k, H = 2, 64, 64, 1, 8
x = recipe.randn((B, ci, H, H), 7); dy = recipe.randn((B, co, H, H), 9)
xp = ops.planes_split_raw(cl(x)); dyp = ops.planes_split_raw(cl(dy))
dw = ops.conv_wgrad_planes_raw(xp, dyp, (co, ci, k, k), k).double().cpu()[:, :, 0, 0]
ref = torch.nn.grad.conv2d_weight(bf(x), (co, ci, k, k), bf(dy), 1, 0)[:, :, 0, 0]
for n in range(20):
    # which ref row (or combination) does dw[n] look like
    d = (ref - dw[n][None, :]).abs().max(1).values
    j = int(d.argmin())
    # column-wise match within ref[n]
    dc = (ref.t() - dw[:, n][None, :]).abs().max(1).values
    print(n, 'closest ref row', j, float(d.min()), '| dw[n,0:4]', [round(float(v), 3) for v in dw[n, :4]], 'ref', [round(float(v), 3) for v in ref[n, :4]])
# try linear combos: is dw[1] = sum of some refs?
A = ref.t().numpy(); 
for n in (1, 2, 3, 5):
    coef, *_ = np.linalg.lstsq(A, dw[n].numpy(), rcond=None)
    big = [(i, round(float(c), 3)) for i, c in enumerate(coef) if abs(c) > 1e-3]
    print('dw[%d] = ' % n, big)
