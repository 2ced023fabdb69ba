"""Diagnostic (Y4_STAMPS build of the library only): per-segment cycle shares of the f16x2 gather kernel's K loop."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, yolov4_amd
from yolov4_amd import ops
yolov4_amd.set_conv_mode('f16x2')
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
def stamps_of(fn, label):
    fn(); torch.cuda.synchronize()
    st = ops._SCRATCH[('cuda', 0)][64:64 + 64].view(torch.int64)
    st.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): fn()
    e1.record(); torch.cuda.synchronize()
    v = st[:5].cpu().tolist(); tot = sum(v[:4])
    print(f'  {label}: {e0.elapsed_time(e1) / 3:.3f} ms; cycles/wave-iter store {v[0]/v[4]:.0f} load {v[1]/v[4]:.0f} compute {v[2]/v[4]:.0f} '
          f'barrier {v[3]/v[4]:.0f} | shares {v[0]/tot:.2f} {v[1]/tot:.2f} {v[2]/tot:.2f} {v[3]/tot:.2f}', flush=True)


print('forward')
for (ci, co, k, H) in [(256, 512, 3, 38), (128, 128, 3, 76), (512, 256, 1, 38), (512, 1024, 3, 19)]:
    x = torch.randn((64, ci, H, H), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((co, ci, k, k), generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    ops.conv_fwd_raw(x, w, k, 1); torch.cuda.synchronize()
    stamps_of(lambda: ops.conv_fwd_raw(x, w, k, 1), f'{ci}->{co} k{k} @{H}')
if os.environ.get('Y4_PROBE_FWD_ONLY'):
    sys.exit(0)
print('dgrad, fp32 dy vs pre-split dy planes')
for (ci, co, k, H) in [(256, 512, 3, 38), (128, 128, 3, 76), (512, 256, 1, 38), (512, 1024, 3, 19)]:
    x = torch.randn((64, ci, H, H), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((co, ci, k, k), generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd_raw(x, w, k, 1)
    mean = y.mean(dim=(0, 2, 3)); invstd = (y.var(dim=(0, 2, 3), unbiased=False) + 1e-5).rsqrt()
    gamma = torch.rand(co, device=dev) + 0.5; beta = torch.zeros(co, device=dev)
    dz = (torch.randn((64, co, H, H), generator=g) * 1e-3).to(dev).contiguous(memory_format=torch.channels_last)
    print(f'{ci}->{co} k{k} @{H}')
    for on in (False, True):
        planes = ops.new_amax(dev, 8) if on else None
        da = planes[5:6] if on else ops.new_amax(dev)
        dy, _, _ = ops.bn_act_bwd_raw(dz, y, mean, invstd, gamma, beta, 'mish', out_amax=None if on else da, planes=planes)
        stamps_of(lambda: ops.conv_dgrad_raw(dy, w, tuple(x.shape), k, 1, dy_amax=da, dy_planes=on), 'planes' if on else 'fp32  ')
