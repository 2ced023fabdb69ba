"""Streaming 1x1 kernel vs the generic path (same library, Y4 eligibility threshold M >= 131072) and vs fp64."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from yolov4_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu'); g.manual_seed(1)
for (B, ci, co, H) in [(3, 64, 64, 211), (2, 32, 64, 300), (2, 64, 128, 270), (2, 64, 32, 301), (2, 32, 32, 260), (2, 64, 96, 270)]:
    x = torch.randn((B, ci, H, H), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((co, ci, 1, 1), generator=g) * 0.1).to(dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((B, co, H, H), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd_raw(x, w, 1, 1)
    ref = torch.nn.functional.conv2d(x.double(), w.double())
    e1 = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    dx = ops.conv_dgrad_raw(dy, w, (B, ci, H, H), 1, 1)
    refd = torch.nn.functional.conv_transpose2d(dy.double(), w.double())
    e2 = (dx.double() - refd).abs().max().item() / refd.abs().max().item()
    # BN statistics path
    rm = torch.zeros(co, device=dev); rv = torch.ones(co, device=dev); nbt = torch.zeros((), dtype=torch.long, device=dev)
    yb, mean, invstd = ops.conv_fwd_bnstats_raw(x, w, 1, 1, rm, rv, nbt, 0.1, 1e-5)
    e3 = (yb.double() - ref).abs().max().item() / ref.abs().max().item()
    mref = ref.mean(dim=(0, 2, 3)); vref = ref.var(dim=(0, 2, 3), unbiased=False)
    e4 = (mean.double() - mref).abs().max().item()
    e5 = ((invstd.double() - (vref + 1e-5).rsqrt()).abs() / (vref + 1e-5).rsqrt()).max().item()
    print(f'B{B} {ci}->{co} @{H}: fwd rel {e1:.2e} dgrad rel {e2:.2e} bnstats-y rel {e3:.2e} mean abs {e4:.2e} invstd rel {e5:.2e}')
