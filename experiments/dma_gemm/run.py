"""Builds and runs the DMA-GEMM prototype (experiments/dma_gemm/proto.hip): correctness vs fp64 on a small case, then
throughput on 1x1-conv-shaped problems next to the product's conv kernel on the same shapes.
usage: python experiments/dma_gemm/run.py"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
R = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, R)
import torch
so = os.path.join(HERE, 'libproto.so')
if not os.path.isfile(so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-shared',
                           os.path.join(HERE, 'proto.hip'), '-o', so])
L = ctypes.CDLL(so)
P, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
L.proto_split.argtypes = [P, P, LL, P]; L.proto_gemm.argtypes = [P, P, P, I, I, I, P]; L.proto_gemm2.argtypes = [P, P, P, I, I, I, P]; L.proto_gemm3.argtypes = [P, P, P, I, I, I, P]
VARIANT = int(os.environ.get('PROTO_VARIANT', '1'))
dev = torch.device('cuda:0')
st = lambda: torch.cuda.current_stream().cuda_stream


def planes(x):
    p = torch.empty((3,) + tuple(x.shape), dtype=torch.int16, device=dev)
    assert L.proto_split(x.data_ptr(), p.data_ptr(), x.numel(), st()) == 0
    return p


_RAW = {}


def planes_or_raw(x):
    if VARIANT == 3:
        return x
    return planes(x)


def gemm(Ap, Bp, M, N, K):
    C = torch.empty((M, N), dtype=torch.float32, device=dev)
    fn = {1: L.proto_gemm, 2: L.proto_gemm2, 3: L.proto_gemm3}[VARIANT]
    rc = fn(Ap.data_ptr(), Bp.data_ptr(), C.data_ptr(), M, N, K, st())
    assert rc == 0, rc
    return C


g = torch.Generator().manual_seed(0)
M, N, K = 512, 256, 160 // 32 * 32
A = torch.randn((M, K), generator=g).to(dev); B = (torch.randn((N, K), generator=g) * 0.1).to(dev)
C = gemm(planes_or_raw(A), planes(B), M, N, K)
ref = A.double() @ B.double().t()
err = float((C.double() - ref).abs().max() / ref.abs().max())
print('check: max err / range =', err)
assert err < 1e-6 or os.environ.get('PROTO_MODE', '0') != '0'

from yolov4_amd import ops
# sustained MFMA-only rate (random bf16 operands in registers, 8 waves per CU, ~50 ms)
L.proto_mfma_peak.argtypes = [P, P, I, P]
seed = (torch.randn(1024, device=dev) * 1.0).view(torch.int32)
out = torch.empty(256 * 512, device=dev)
for iters in (2000, 20000, 20000):
    L.proto_mfma_peak(seed.data_ptr(), out.data_ptr(), iters, st()); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.proto_mfma_peak(seed.data_ptr(), out.data_ptr(), iters, st()); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1)
    fl = 256 * 8 * iters * 16 * 2.0 * 32 * 32 * 16
    print(f'MFMA-only loop: {iters} iters {t:8.2f} ms -> {fl / t / 1e9:7.0f} TFLOP/s bf16 sustained', flush=True)
# full-machine shapes first (256 or 512 blocks of 256 x 256: no tile quantisation), then the layer shapes
for (M, K, N) in [(65536, 512, 256), (65536, 4608, 256), (131072, 4608, 256), (65536, 2304, 512)]:
    A = torch.randn((M, K), device=dev); B = torch.randn((N, K), device=dev) * 0.05
    Ap, Bp = planes_or_raw(A), planes(B)
    gemm(Ap, Bp, M, N, K); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gemm(Ap, Bp, M, N, K)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    fl = 2.0 * M * N * K
    print(f'M={M:7d} K={K:5d} N={N:4d}: DMA planes GEMM {t:7.3f} ms {fl / t / 1e9:6.1f} TF (x6 = {6 * fl / t / 1e9:6.0f} TF bf16)  [{M // 256 * (N // 256)} blocks]', flush=True)
    del A, B, Ap, Bp
for (H, K, N) in [(38, 512, 256), (19, 1024, 512), (38, 256, 256), (19, 512, 512), (19, 4608, 1024)]:
    M = 64 * H * H // 256 * 256
    A = torch.randn((M, K), device=dev); B = torch.randn((N, K), device=dev) * 0.05
    Ap, Bp = planes_or_raw(A), planes(B)
    gemm(Ap, Bp, M, N, K); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gemm(Ap, Bp, M, N, K)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    x = torch.randn((64, K, H, H), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((N, K, 1, 1), device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    ops.conv_fwd_raw(x, w, 1, 1); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): ops.conv_fwd_raw(x, w, 1, 1)
    e1.record(); torch.cuda.synchronize()
    t2 = e0.elapsed_time(e1) / 10
    fl = 2.0 * M * N * K
    print(f'M={M:7d} K={K:5d} N={N:4d}: DMA planes GEMM {t:7.3f} ms {fl / t / 1e9:6.1f} TF (x6 = {6 * fl / t / 1e9:6.0f} TF bf16) | '
          f'product conv kernel {t2:7.3f} ms {2.0 * 64 * H * H * N * K / t2 / 1e9:6.1f} TF', flush=True)
