"""Accuracy of the two conv arithmetic modes against an fp64 CPU convolution."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, torch.nn.functional as F
import yolov4_amd
from yolov4_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for (B, ci, co, k, H) in [(2, 512, 256, 3, 19), (2, 128, 128, 3, 38), (4, 1024, 512, 1, 19)]:
    x = torch.randn((B, ci, H, H), generator=g); w = torch.randn((co, ci, k, k), generator=g) / (ci * k * k) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, 1, (k - 1) // 2)
    cpu32 = F.conv2d(x, w, None, 1, (k - 1) // 2).double()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last); wd = w.to(dev).contiguous(memory_format=torch.channels_last)
    out = {}
    for mode in (0, 1):
        yolov4_amd.set_conv_mode(mode)
        out[mode] = ops.conv_fwd_raw(xd, wd, k, 1).double().cpu()
    s = ref.abs().max()
    def err(t): return float((t - ref).abs().max() / s), float(((t - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    print(f'K={ci*k*k:5d}: cpu-fp32 max/rms {err(cpu32)[0]:.2e}/{err(cpu32)[1]:.2e} | hip fp32-mfma {err(out[0])[0]:.2e}/{err(out[0])[1]:.2e} | hip bf16x3 {err(out[1])[0]:.2e}/{err(out[1])[1]:.2e}')
