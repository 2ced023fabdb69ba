"""Micro-benchmark of the BatchNorm+activation sweeps (HBM-bound): achieved TB/s per kernel family.
usage: python scripts/bn_micro.py [reps]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from yolov4_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
B = 64
for (C, H) in [(64, 304), (128, 152), (256, 76), (512, 38), (1024, 19), (32, 608)]:
    y = torch.randn((B, C, H, H), device=dev).contiguous(memory_format=torch.channels_last)
    dz = torch.randn_like(y)
    g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev) * 0.1
    mean, invstd = ops.bn_stats_raw(y, None, None, None, 0.1, 1e-5)
    nbytes = y.numel() * 4
    def timeit(fn):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_f = timeit(lambda: ops.bn_act_fwd_raw(y, mean, invstd, g, b, 'mish'))
    t_b = timeit(lambda: ops.bn_act_bwd_raw(dz, y, mean, invstd, g, b, 'mish'))
    t_c = timeit(lambda: y.clone())
    t_s = timeit(lambda: ops.bn_stats_raw(y, None, None, None, 0.1, 1e-5))
    print(f'C={C:5d} H={H:3d} {nbytes/1e6:7.0f} MB | fwd {t_f:6.3f} ms {2*nbytes/t_f/1e9:5.2f} TB/s | bwd(reduce+apply) {t_b:6.3f} ms {5*nbytes/t_b/1e9:5.2f} TB/s | stats {t_s:6.3f} ms {nbytes/t_s/1e9:5.2f} TB/s | torch clone {2*nbytes/t_c/1e9:5.2f} TB/s', flush=True)
    del y, dz
