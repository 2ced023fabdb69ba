"""dy-planes experiment: BatchNorm backward writes dy pre-split (fp16 hi/lo planes); dgrad and wgrad consume them.
Checks both against the fp32-dy path and times BN-bwd + dgrad + wgrad either way."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, yolov4_amd
from yolov4_amd import ops
yolov4_amd.set_conv_mode('f16x2')
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (ci, co, k, H, B) in [(256, 512, 3, 38, 64), (128, 128, 3, 76, 64), (512, 256, 1, 38, 64), (512, 1024, 3, 19, 64),
                          (64, 64, 1, 152, 32), (64, 64, 3, 152, 32), (1024, 512, 1, 19, 64)]:
    x = torch.randn((B, ci, H, H), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((co, ci, k, k), generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd_raw(x, w, k, 1)
    mean = y.mean(dim=(0, 2, 3)); var = y.var(dim=(0, 2, 3), unbiased=False); invstd = (var + 1e-5).rsqrt()
    gamma = torch.rand(co, device=dev) + 0.5; beta = torch.randn(co, device=dev) * 0.1
    dz = (torch.randn((B, co, H, H), generator=g) * 1e-3).to(dev).contiguous(memory_format=torch.channels_last)
    xa = ops.amax_raw(x)

    def path(planes_on):
        planes = ops.new_amax(dev, 8) if planes_on else None
        da = planes[5:6] if planes_on else ops.new_amax(dev)
        dy, dg, db = ops.bn_act_bwd_raw(dz, y, mean, invstd, gamma, beta, 'mish', out_amax=None if planes_on else da, planes=planes)
        dx = ops.conv_dgrad_raw(dy, w, tuple(x.shape), k, 1, dy_amax=da, dy_planes=planes_on)
        dw = ops.conv_wgrad_raw(x, dy, tuple(w.shape), k, 1, x_amax=xa, dy_amax=da, dy_planes=planes_on)
        return dx, dw, dg, db
    a = path(False); b = path(True)
    def rel(u, v): return ((u - v).double().norm() / v.double().norm()).item()
    t0 = timed(lambda: path(False)); t1 = timed(lambda: path(True))
    print(f'{ci}->{co} k{k} @{H} B{B}: rel dx {rel(b[0], a[0]):.2e} dw {rel(b[1], a[1]):.2e} dgamma {rel(b[2], a[2]):.1e} | '
          f'fp32-dy {t0:.3f} ms  planes {t1:.3f} ms  ({(t0 - t1) / t0 * 100:+.1f}%)', flush=True)
