#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Headline benchmark (BASELINE.json): images/sec of one YOLOv4 training step
(forward + YOLOLoss + backward, no optimizer) at 608x608, batch 64 per GPU,
synthetic inputs per SURVEY.md §8(d) config 3, fp32-grade conv arithmetic (f16x2 by default; --conv-mode bf16x3 / f32
for the 3-piece bf16 split / the fp32 MFMA).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is measured live: every launch of the
convolution kernels inside the timed steps is bracketed by HIP events on the
launch stream and attributed to the kernel symbol the dispatcher picks; achieved = algorithmic conv FLOPs of the
launches of the ONE symbol with the largest total time / their summed durations (`per_symbol` lists all of them).  `cpu_baseline` is
the oracle (torch CPU fp32 restatement of the reference) running the same step
on the host cores at a bounded batch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

# the pool's host driver only supports dmabuf IPC (RCCL / cross-process tensor sharing); already exported there
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np
import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense
FLOP_PER_IMG_608 = 402.63e9       # fwd + dgrad + wgrad conv flops per image @608


class _QuietStdout:
    """stdout carries exactly ONE line, the JSON result: while the job runs, file descriptor 1 points at stderr, so that whatever
    a library prints there (RCCL announces its version on stdout when the first communicator is created) cannot get in
    front of it; `emit` restores the descriptor and prints the line."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        print(line, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU')
    ap.add_argument('--size', type=int, default=608)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=1)
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--all-kernel-events', action='store_true',
                    help='HIP-event brackets around EVERY conv launch (per_symbol / all_conv_kernels cover all of them); default: '
                         'only around the launches of the dominant kernel, conv_planes_mfma -- an event pair costs ~10 us of '
                         'stream time, 3 ms per step over all ~330 conv launches (implied by --conv-table)')
    ap.add_argument('--ddp-timeline', action='store_true',
                    help='record when each gradient bucket becomes ready inside the backward pass (adds ddp_timeline to the line)')
    ap.add_argument('--conv-mode', default='f16x2', choices=['f16x2', 'bf16x3', 'f32', 'bf16', 'bf16_all'],
                    help='conv arithmetic: 2-piece fp16 split (fp32-grade, 3 MFMAs per product, default), exact 3-way bf16 '
                         'split (6 MFMAs), the fp32 MFMA fma chain, or plain bf16 operands (mixed precision)')
    ap.add_argument('--conv-table', default=None, help='write a per-shape conv timing table to this file')
    ap.add_argument('--infer', action='store_true',
                    help='BASELINE configs[1] instead of the headline: eval forward + YOLO decode + postprocess/NMS at '
                         '--size, --batch 32 by default; reports images/sec and achieved GB/s of the decode / NMS kernels')
    ap.add_argument('--no-infer-leg', action='store_true',
                    help='skip the configs[1] inference leg that the default N = 1 run attaches as `inference` after the timed region')
    ap.add_argument('--stub-step', action='store_true',
                    help='TEST ONLY (tests/test_host_logic.py): a CPU / gloo stand-in for the step, to exercise the launcher, '
                         'barrier and max-over-ranks plumbing without a GPU; the line it prints is marked "stub": true and is not a measurement')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (the form the driver uses for N = 1): start the N ranks
    as CHILD processes through torch.distributed.run -- this parent has not touched the GPU (importing torch does not), and
    it never execs -- relay rank 0's one JSON line and exit with the workers' status (reference launch: main_amp.py:94-98,126-131
    under `python -m torch.distributed.launch`)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, Y4_BENCH_CHILD='1')
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [l for l in proc.stdout.decode(errors='replace').splitlines() if l.startswith('{')]
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if proc.returncode else (0 if lines else 1)


def stub_main(args):
    """--stub-step: the distributed skeleton of main() on CPU tensors over gloo (same rendezvous variables, barrier + timing
    + MAX over ranks + one line from rank 0), nothing of the hot path."""
    quiet = _QuietStdout()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if world > 1:
        dist.init_process_group('gloo')
    w = torch.zeros(1024)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g = torch.full((1024,), float(rank + 1))
        if world > 1:
            dist.all_reduce(g)
            g /= world
        w += g
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        quiet.emit(json.dumps({'stub': True, 'metric': 'launcher plumbing test, not a measurement', 'value': 0.0, 'n_gpus': world,
                               'steps': args.steps, 'warmup': args.warmup, 'mean_grad': float(w[0]) / max(args.steps, 1),
                               'seconds_max_over_ranks': float(t.item())}))
    if world > 1:
        dist.destroy_process_group()


class ConvTimer:
    """HIP-event brackets around the conv launches (same stream as the kernels)."""

    def __init__(self):
        self.rec = []
        self.on = False
        self.sym = None

    def wrap(self, ops, only_dominant=False):
        timer = self

        def bracket(fn, kind, flops_of):
            def inner(*a, **kw):
                if not timer.on:
                    return fn(*a, **kw)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **kw)
                e1.record()
                fl, key = flops_of(*a, **kw)
                # the library names the kernel it launched (f16x2 mode); other modes: the dispatch rules restated below
                timer.rec.append((kind, fl, e0, e1, key, ops.last_conv_kernel() or timer.sym))
                return out
            return inner

        def f_fwd(x, w, k, s, *a, **kw):
            B, Cin, H, W = x.shape
            Ho, Wo = ops.conv_out_hw(H, W, k, s)
            timer.sym = gather_symbol(False, B * Ho * Wo, w.shape[0], Cin, Cin, k, s, ops.nhwc_pitch(x) if Cin != 3 else 0)
            return 2.0 * B * Ho * Wo * w.shape[0] * Cin * k * k, (Cin, w.shape[0], k, s, H)

        def f_dgrad(dy, w, x_shape, k, s, *a, **kw):
            B, Cout, Ho, Wo = dy.shape
            cpad = (Cout + 31) // 32 * 32
            timer.sym = gather_symbol(True, x_shape[0] * x_shape[2] * x_shape[3], x_shape[1], cpad, Cout, k, s,
                                      max(ops.nhwc_pitch(dy), cpad))
            return 2.0 * B * Ho * Wo * Cout * x_shape[1] * k * k, (x_shape[1], Cout, k, s, x_shape[2])

        def f_wgrad(x, dy, w_shape, k, s, *a, **kw):
            B, Cout, Ho, Wo = dy.shape
            mode = args_conv_mode()
            kern = {'f32': 'conv_wgrad_mfma_f32', 'f16x2': 'conv_wgrad_f16x2'}.get(mode, 'conv_wgrad_bf16x3')
            tn, tj = (64 if Cout <= 64 else 128), (64 if k * k * x.shape[1] <= 64 else 128)
            tail = {'f32': '>', 'bf16': ', 1>', 'bf16x3': ', 3>', 'f16x2': f", {os.environ.get('Y4_F16X2_SHAPE', '16')}>"}[mode]
            timer.sym = 'conv_stem_wgrad_kernel' if x.shape[1] == 3 else f'{kern}<{tn}, {tj}' + tail
            return 2.0 * B * Ho * Wo * Cout * x.shape[1] * k * k, (x.shape[1], Cout, k, s, x.shape[2])

        # DMA-fed kernels over pre-split operands (conv_planes.hip): x / dy arrive as ops.Planes, stride 1 in backward
        def f_fwd_p(xp, w, k, s, *a, **kw):
            B, Cin, H, W = xp.shape
            Ho, Wo = ops.conv_out_hw(H, W, k, s)
            return 2.0 * B * Ho * Wo * w.shape[0] * Cin * k * k, (Cin, w.shape[0], k, s, H)

        def f_dgrad_p(dyp, w, x_shape, k, *a, **kw):
            B, Cout, Ho, Wo = dyp.shape
            return 2.0 * B * Ho * Wo * Cout * x_shape[1] * k * k, (x_shape[1], Cout, k, 1, x_shape[2])

        def f_wgrad_p(xp, dyp, w_shape, k, *a, **kw):
            B, Cout, Ho, Wo = dyp.shape
            return 2.0 * B * Ho * Wo * Cout * xp.shape[1] * k * k, (xp.shape[1], Cout, k, 1, xp.shape[2])

        ops.conv_fwd_planes_bnstats_raw = bracket(ops.conv_fwd_planes_bnstats_raw, 'conv_fwd', f_fwd_p)
        ops.conv_dgrad_planes_raw = bracket(ops.conv_dgrad_planes_raw, 'conv_dgrad', f_dgrad_p)
        if only_dominant:
            return
        ops.conv_wgrad_planes_raw = bracket(ops.conv_wgrad_planes_raw, 'conv_wgrad', f_wgrad_p)
        ops.conv_fwd_raw = bracket(ops.conv_fwd_raw, 'conv_fwd', f_fwd)
        ops.conv_fwd_bnstats_raw = bracket(ops.conv_fwd_bnstats_raw, 'conv_fwd', f_fwd)   # conv + BN-stat epilogue + fold
        ops.conv_dgrad_raw = bracket(ops.conv_dgrad_raw, 'conv_dgrad', f_dgrad)
        ops.conv_wgrad_raw = bracket(ops.conv_wgrad_raw, 'conv_wgrad', f_wgrad)

    def by_symbol(self):
        """Seconds / flop / launches per kernel symbol (the names rocprofv3 --stats prints)."""
        agg = {}
        for kind, fl, e0, e1, key, sym in self.rec:
            a = agg.setdefault(sym, [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        return {k: {'tflops': v[0] / v[1] / 1e12, 'seconds': v[1], 'launches': v[2]} for k, v in agg.items()}

    def summary(self):
        agg = {}
        for kind, fl, e0, e1, _key, _sym in self.rec:
            a = agg.setdefault(kind, [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        return {k: {'tflops': v[0] / v[1] / 1e12, 'seconds': v[1], 'launches': v[2], 'flop': v[0]} for k, v in agg.items()}

    def table(self, steps, batch=64):
        """Per layer shape and kernel: launches and ms per step, TFLOP/s, and TB/s over the unavoidable bytes (input + output
        tensor once each, 4 B per element) -- the second tells which rows are HBM-bound rather than MFMA-bound."""
        agg = {}
        for kind, fl, e0, e1, key, sym in self.rec:
            a = agg.setdefault((kind,) + key + ((sym or '?').split('(')[0],), [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        lines = ['kind        Cin  Cout k s  Hin   n/step  ms/step  TFLOP/s   TB/s  kernel']
        for (kind, cin, cout, k, s, h, sym), v in rows:
            ho = (h + 2 * ((k - 1) // 2) - k) // s + 1
            nbytes = 4.0 * batch * (h * h * cin + ho * ho * cout) * v[2]
            lines.append(f'{kind:10s} {cin:5d} {cout:5d} {k} {s} {h:4d} {v[2] / steps:7.1f} {v[1] / steps * 1e3:8.2f} {v[0] / v[1] / 1e12:8.1f} '
                         f'{nbytes / v[1] / 1e12:6.2f}  {sym}')
        return '\n'.join(lines)


def gather_symbol(transposed, M, N, Cs, Cs_valid, k, stride, ld):
    """The kernel symbol dispatch_gather (yolov4_amd/csrc/conv_igemm.hip) launches for this problem -- the same rules,
    restated so that the roofline block can name ONE kernel as rocprofv3 --stats names it.  M, N = GEMM rows/columns,
    Cs = channels of the gathered tensor (padded), ld = its pixel pitch."""
    mode = args_conv_mode()
    if Cs == 3:
        return 'conv_stem_fwd_bf16x3_kernel' if mode in ('bf16x3', 'f16x2', 'bf16') else 'conv_stem_fwd_kernel'
    tr = 'true' if transposed else 'false'
    if mode == 'f16x2':
        ms = int(os.environ.get('Y4_F16X2_SHAPE', '16'))
        if (k == 1 and stride == 1 and Cs in (32, 64, 128) and Cs_valid == Cs and N <= 128 and M * ld * 4 < 0xfffffff0
                and M >= 128 * 1024):
            nt = (N + 31) // 32
            nt = 4 if nt >= 3 else nt
            return f'conv1x1_stream_f16x2<{Cs // 16}, {nt}, {8 if (Cs == 128 and nt == 4) else 4}>'
        if N > 64:
            nt = (N + 127) // 128
            b128, b64 = (M + 127) // 128 * nt, (M + 63) // 64 * nt
            c128 = ((b128 + 511) // 512) * 128.0
            c64 = ((b64 + 511) // 512) * 64.0 * 1.10
            bm = 64 if (c64 < c128 and not (transposed and stride == 2)) else 128
            return f'conv_gather_f16x2<{bm}, 128, 2, 2, {tr}, {ms}>'
        if N > 32:
            return f'conv_gather_f16x2<128, 64, 2, 2, {tr}, {ms}>'
        return f'conv_gather_f16x2<128, 32, 4, 1, {tr}, {ms}>'
    if (mode == 'bf16x3' and k == 1 and stride == 1 and Cs in (32, 64, 128) and Cs_valid == Cs and N <= 128
            and Cs * ((N + 31) // 32 * 32) <= (16384 if Cs == 128 else 8192) and M * ld * 4 < 0xfffffff0 and M >= 128 * 1024):
        nt = (N + 31) // 32
        nt = 4 if nt >= 3 else nt
        return f'conv1x1_stream_bf16x3<{Cs // 16}, {nt}, {8 if (Cs == 128 and nt == 4) else 4}>'
    split = mode != 'f32'
    kern = 'conv_gather_bf16x3' if split else 'conv_gather_mfma_f32'
    tail = (', 1>' if mode == 'bf16' else ', 3>') if split else None
    def name(bm, bn, wm, wn, bk=32):
        return f'{kern}<{bm}, {bn}, {wm}, {wn}, {tr}' + (tail if split else f', {bk}>')
    if N > 64:
        nt = (N + 127) // 128
        b128, b64 = (M + 127) // 128 * nt, (M + 63) // 64 * nt
        slots = 512 if split or k != 1 else 768
        c128 = ((b128 + slots - 1) // slots) * 128.0
        c64 = ((b64 + slots - 1) // slots) * 64.0 * (1.10 if split else 1.08)
        bk = 16 if (not split and k == 1) else 32
        if c64 < c128 and not (transposed and stride == 2):
            return name(64, 128, 2, 2, bk)
        return name(128, 128, 2, 2, bk)
    if N > 32:
        return name(128, 64, 2, 2, 16 if (not split and k == 1) else 32)
    return name(128, 32, 4, 1)


MODES = {
    'f16x2': {'kk': 'f16x2', 'mfmas': 3, 'peak': PEAK_BF16_MFMA_TFLOPS, 'instr': 'v_mfma_f32_32x32x16_f16 / 16x16x32_f16',
              'peak_note': 'dense fp16 MFMA peak 2500 / 3 MFMAs per fp32-grade product',
              'text': 'f16x2: fp32 operands scaled by a per-tensor power of two and split into 2 fp16 pieces (11 + 11 bits), '
                      '3 fp16 MFMAs per product, two fp32 accumulators (per-product error ~2^-22: rms 4.3e-7 of the output at '
                      'K = 4608 vs 1.19e-6 for an fp32 fma chain, tests/test_gpu_parity.py::test_conv_modes_accuracy_vs_fp64)'},
    'bf16x3': {'kk': 'bf16x3', 'mfmas': 6, 'peak': PEAK_BF16_MFMA_TFLOPS, 'instr': 'v_mfma_f32_32x32x16_bf16',
               'peak_note': 'dense bf16 MFMA peak 2500 / 6 MFMAs per fp32-exact product',
               'text': 'bf16x3: fp32 operands split exactly into 3 bf16 pieces, 6 bf16 MFMAs per product, fp32 accumulate '
                       '(error <= the fp32-MFMA fma chain, see DESIGN.md)'},
    'bf16': {'kk': 'f16x2', 'mfmas': 1, 'peak': PEAK_BF16_MFMA_TFLOPS, 'instr': 'v_mfma_f32_16x16x32_bf16',
             'peak_note': 'dense bf16 MFMA peak (the dominant kernel runs ONE bf16 MFMA per product)',
             'text': 'BASELINE configs[4], mixed precision: the layers on the DMA-fed plane kernels (72 of the 107 BatchNorm layers, '
                     '9/10 of the conv flops) take plain bf16 operands (RN, written by the BatchNorm sweeps) and run one bf16 MFMA per '
                     'product with fp32 accumulation; the remaining, HBM-bound layers keep the fp32-grade f16x2 kernels; BatchNorm '
                     'statistics, loss and NMS in fp32 (yolov4_amd.set_conv_mode("bf16") = conv mode 3 + y4_set_planes_bf16)'},
    'bf16_all': {'kk': 'bf16x3', 'mfmas': 1, 'peak': PEAK_BF16_MFMA_TFLOPS, 'instr': 'v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16',
                 'peak_note': 'dense bf16 MFMA peak',
                 'text': 'every conv operand rounded to bf16 (RN), one bf16 MFMA per product, fp32 accumulate; fp32 BN/loss/NMS '
                         '(conv mode 2: plane layers on the bf16 DMA kernels, the others on the register-staged bf16 kernels)'},
    'f32': {'kk': 'mfma_f32', 'mfmas': 1, 'peak': PEAK_F32_MFMA_TFLOPS, 'instr': 'v_mfma_f32_32x32x2_f32',
            'peak_note': 'dense fp32 MFMA peak', 'text': 'fp32 MFMA fma chain (v_mfma_f32_32x32x2_f32)'},
}


def pmc_recorded(kind, sym):
    """(value, source file) for one kernel symbol from the newest committed rocprofv3 --pmc recording of this command
    under profiles/ (kind 'hbm_traffic': bytes per launch, 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes;
    kind 'mfma_util': SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed cycles)).  Recorded, not live."""
    import glob
    key = {'hbm_traffic': 'hbm_bytes_per_launch_corrected', 'mfma_util': 'mfma_util'}[kind]
    pat = {'hbm_traffic': 'r*_pmc_hbm_traffic_per_kernel*.json', 'mfma_util': 'r*_pmc_mfma_util_per_kernel*.json'}[kind]
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', pat)), reverse=True):
        try:
            for k, v in json.load(open(path)).items():
                if sym in k and key in v:
                    return v[key], os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None, None


_ARGS = {}


def args_conv_mode():
    return _ARGS.get('conv_mode', 'f16x2')


def cpu_baseline(size, batch):
    import recipe
    from oracle import network as NW
    threads = torch.get_num_threads()
    sd = NW.empty_state_dict()
    recipe.fill_state_dict_(sd, 1234)
    net = NW.RefNet(sd, recipe.MODEL_CFG)
    x = recipe.randn((batch, 3, size, size), 5)
    labels = recipe.synth_labels(batch, size, 6).numpy()
    # the metric's own unit first (train step fwd+loss+bwd): 1 warm-up + 2 timed runs, best of the two
    net.train_step(x, labels)
    dts = []
    for _ in range(2):
        t0 = time.time()
        net.train_step(x, labels)
        dts.append(time.time() - t0)
    dt = min(dts)
    # and the reference's val.py path: eval forward, 1 warm-up + median of 3
    with torch.no_grad():
        net.forward_eval(x)
        fts = []
        for _ in range(3):
            t0 = time.time()
            net.forward_eval(x)
            fts.append(time.time() - t0)
    ft = sorted(fts)[1]
    # configs[0]: the reference's own CPU-runnable case, eval forward of 1x3x416x416
    x0 = recipe.randn((1, 3, 416, 416), 7)
    with torch.no_grad():
        net.forward_eval(x0)
        f0 = []
        for _ in range(5):
            t0 = time.time()
            net.forward_eval(x0)
            f0.append(time.time() - t0)
    f0 = sorted(f0)[2]
    return {'value': batch / dt, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'forward_only_images_per_sec': batch / ft,
            'config0_416_forward_images_per_sec': 1.0 / f0,
            'sample': f'oracle (torch-CPU fp32 restatement of the reference) on {batch}x3x{size}x{size}: train step '
                      f'fwd+loss+bwd best of 2 after 1 warm-up = {dt:.2f} s; eval forward median of 3 = {ft:.2f} s; '
                      f'configs[0] eval forward 1x3x416x416 median of 5 = {f0:.3f} s; '
                      f'torch threads {threads}, host cpus {os.cpu_count()}'}


def infer_leg(args, steps=None, warmup=None):
    """BASELINE configs[1]: 1xMI355X inference, 608x608 bs=32 (val.py path: eval forward -> postprocess).  One JSON line:
    images/sec of forward + postprocess, and for the HBM-bound head kernels the achieved GB/s against their algorithmic
    bytes (decode reads + writes B*N*85*4 B = 15.47 MB/img; the candidate count reads it once more), measured
    live with HIP events on the launch stream."""
    import recipe
    import yolov4_amd
    from yolov4_amd import ops
    from yolov4_amd._lib import lib
    from yolov4_amd.yolo.model.yolov4 import YOLOv4
    from yolov4_amd.yolo.util import utils as U
    yolov4_amd.set_conv_mode(args.conv_mode)
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(dev)
    B, S = (args.batch if args.batch != 64 else 32), args.size
    steps = steps if steps is not None else args.steps
    warmup = warmup if warmup is not None else args.warmup
    m = YOLOv4(recipe.MODEL_CFG, device=dev)
    sd = m.state_dict()
    recipe.fill_state_dict_(sd, 1234)
    m.load_state_dict(sd)
    m = m.to(dev)
    recipe.calibrate_bn_(m, recipe.randn((8, 3, S, S), 77).to(dev))      # random weights need statistics at this size
    m.eval()
    x = recipe.randn((B, 3, S, S), 78).to(dev)
    rec = {}

    def bracket(obj, name, key):
        fn = getattr(obj, name)

        def inner(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            rec.setdefault(key, []).append((e0, e1))
            return out
        setattr(obj, name, inner)

    bracket(ops, 'yolo_decode_eval', 'decode')
    L = lib()
    bracket(L, 'y4_post_count_f32', 'post_count')
    bracket(L, 'y4_post_nms_f32', 'post_fill_sort_nms')
    with torch.no_grad():
        for _ in range(max(warmup, 1)):
            out = m(x)
        sc = (out[..., 4:5] * out[..., 5:]).flatten()
        # confidence threshold giving ~500 candidates per image (survivors 1e2..1e3 per image)
        thr = float(torch.quantile(sc[torch.randint(0, sc.numel(), (1000000,), device=dev)], 1 - 500.0 / (out.shape[1] * 80)))
        U.postprocess(out.clone(), 80, thr, 0.4)
        rec.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tf = tp = 0.0
        for _ in range(steps):
            a = time.perf_counter()
            out = m(x)
            torch.cuda.synchronize()
            b_ = time.perf_counter()
            det = U.postprocess(out, 80, thr, 0.4)
            torch.cuda.synchronize()
            tf += b_ - a
            tp += time.perf_counter() - b_
        dt = time.perf_counter() - t0
    N = out.shape[1]
    ms = lambda key: sum(e0.elapsed_time(e1) for e0, e1 in rec[key]) / steps
    dec_bytes = 2.0 * B * N * 85 * 4
    cnt_bytes = 1.0 * B * N * 85 * 4 + B * N * 4 * 4        # reads every score once, rewrites the 4 box columns (xyxy)
    surv = sum(0 if d is None else len(d) for d in det) / B
    # the three decode launches of a batch in call order: F = S/8, S/16, S/32
    per_layer = {}
    for li, F_ in enumerate((S // 8, S // 16, S // 32)):
        evs = rec['decode'][li::3]
        t_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(len(evs), 1)
        by = 2.0 * B * 3 * F_ * F_ * 85 * 4
        per_layer[f'F={F_}'] = {'ms': t_ms, 'GB/s': by / (t_ms * 1e-3) / 1e9, 'algorithmic_bytes': by}
    outj = {'metric': f'images/sec inference @{S}x{S} bs={B} (eval forward + YOLO decode + postprocess / per-class NMS)',
            'value': B * steps / dt, 'unit': 'images/sec', 'n_gpus': 1, 'steps': steps, 'warmup': warmup,
            'ms_per_step': dt / steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'configs[1]: 1xMI355X inference, {S}x{S} bs={B}, BN folded into the conv epilogues, random '
                                   f'weights with BatchNorm statistics calibrated at {S}', 'conv_arithmetic': MODES[args.conv_mode]['text'],
                       'conf_thre': thr, 'nms_thre': 0.4, 'survivors_per_img': surv, 'boxes_per_img': N,
                       'forward_ms': tf / steps * 1e3, 'postprocess_ms': tp / steps * 1e3,
                       'forward_images_per_sec': B * steps / tf,
                       'forward_conv_tflops': B * steps / tf * 134.422e9 * (S / 608.0) ** 2 / 1e12},
            'roofline': {'bound': 'hbm', 'kernel': 'yolo_decode_tiled_kernel<true, 32 | 16> (3 launches per batch)',
                         'achieved': dec_bytes / (ms('decode') * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': dec_bytes / (ms('decode') * 1e-3) / 1e9 / 8000.0, 'traffic': None,
                         'algorithmic_bytes_per_batch': dec_bytes, 'ms_per_batch': ms('decode'),
                         'measured': 'LIVE: HIP events around the three decode launches of every timed batch',
                         'per_layer': per_layer,
                         'other_kernels': {
                             'post_count_tiled_kernel (xywh->xyxy + candidate count)': {
                                 'ms_per_batch': ms('post_count'), 'GB/s': cnt_bytes / (ms('post_count') * 1e-3) / 1e9,
                                 'algorithmic_bytes_per_batch': cnt_bytes},
                             'post_scan + post_fill + segment sort + post_nms (data dependent)': {
                                 'ms_per_batch': ms('post_fill_sort_nms'), 'candidates_per_img_target': 500,
                                 'survivors_per_img': surv}}}}
    return outj


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    if args.stub_step:
        return stub_main(args)
    if args.infer:
        quiet = _QuietStdout()
        return quiet.emit(json.dumps(infer_leg(args)))
    quiet = _QuietStdout()
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dist = world > 1 or os.environ.get('Y4_FORCE_DIST') == '1'     # the latter: rehearse the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)

    import recipe
    from yolov4_amd import ops
    from yolov4_amd.ddp import BucketedDDP
    from yolov4_amd.yolo.model.yolov4 import YOLOv4
    from yolov4_amd.yolo.model.yololoss import YOLOLoss

    import yolov4_amd
    yolov4_amd.set_conv_mode(args.conv_mode)
    # (kernel names the library does not report itself are restated from the dispatch rules of the mode the non-plane layers run in)
    _ARGS['conv_mode'] = {'bf16': 'f16x2', 'bf16_all': 'bf16'}.get(args.conv_mode, args.conv_mode)
    timer = ConvTimer()
    all_events = args.all_kernel_events or bool(args.conv_table) or args.conv_mode not in ('f16x2', 'bf16') or not ops.PLANES['on']
    if not args.no_kernel_events:
        timer.wrap(ops, only_dominant=not all_events)

    cfg = recipe.MODEL_CFG
    torch.manual_seed(0)
    model = YOLOv4(cfg, device=dev)
    sd = model.state_dict()
    recipe.fill_state_dict_(sd, 1234)                # identical initial weights on every rank
    model.load_state_dict(sd)
    model = model.to(dev).train()
    ddp = BucketedDDP(model, bucket_mb=25.0)
    if args.ddp_timeline:
        ddp.record_timeline()
    crit = YOLOLoss(cfg, ignore_thresh=0.7, device=dev)

    B, S = args.batch, args.size
    g = torch.Generator(device='cpu')
    g.manual_seed(1000 + rank)                       # per-rank shard of the synthetic global batch
    x = torch.randn((B, 3, S, S), generator=g).to(dev)
    labels = recipe.synth_labels(B, S, 2000 + rank).to(dev)

    from yolov4_amd import trace

    def step():
        ddp.zero_grad()
        with trace.range('y4.forward'):                  # roctx ranges with Y4_ROCTX=1 (no-ops otherwise)
            out_ = ddp(x)
        with trace.range('y4.loss'):
            loss = crit(out_, {'padded_labels': labels})
        with trace.range('y4.backward'):
            loss.backward()                              # the exchange is waited for by BucketedDDP's end-of-backward callback
        return loss

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    timer.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    timer.on = False
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    lossv = float(loss.detach())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        flop_img = FLOP_PER_IMG_608 * (S / 608.0) ** 2
        out = {
            'metric': f'images/sec fwd+bwd @{S}x{S} bs={B} per GPU (YOLOv4 training step: forward + YOLOLoss + backward)',
            'value': value, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if args.conv_mode in ('bf16', 'bf16_all') else 'f32', 'data': 'synthetic',
            'config': {'workload': f'configs[2]: 1xMI355X training step, {S}x{S} bs={B}/GPU, fwd+bwd+YOLOLoss HIP kernels, '
                                   f'synthetic targets (SURVEY 8d config 3); random-init weights',
                       'global_batch': world * B, 'img_size': S, 'parallelism': f'dp{world}',
                       'conv_arithmetic': MODES[args.conv_mode]['text'],
                       'peak_hbm_gib': round(torch.cuda.max_memory_allocated(dev) / 2**30, 1),
                       'loss': lossv, 'conv_tflops_whole_step': value / world * flop_img / 1e12},
        }
        if args.ddp_timeline:
            tl = ddp.timeline()           # of the last timed step
            out['ddp_timeline'] = {
                'note': 'per gradient bucket of the LAST step: MiB, ms after the first gradient hook of the backward pass at which '
                        'its last gradient has landed (= earliest start of its all-reduce), ms of backward work still to run after that',
                'buckets': [{'bucket': i, 'mib': round(nb / 2**20, 2), 'ready_ms': round(a, 2), 'backward_left_ms': round(b, 2)}
                            for i, nb, a, b in tl]}
        if timer.rec:
            summ = timer.summary()
            syms = timer.by_symbol()
            dsym = max(syms, key=lambda k: syms[k]['seconds'])       # ONE kernel symbol, as rocprofv3 --stats names it
            fam_of = {'conv_wgrad': 'conv_wgrad', 'conv_stem_wgrad': 'conv_wgrad', 'wgrad_planes': 'wgrad_planes',
                      'conv_planes': 'conv_planes'}
            dom = next((f for p_, f in fam_of.items() if dsym.startswith(p_)), None) or \
                ('conv_dgrad' if ', true' in dsym else 'conv_fwd')
            mi = MODES[args.conv_mode]
            kname = {'conv_fwd': f"conv_gather_{mi['kk']}<..,false> (forward implicit GEMM; filter split, BN-stat fold kernels included)",
                     'conv_dgrad': f"conv_gather_{mi['kk']}<..,true> (dgrad implicit GEMM; filter transpose/split included)",
                     'conv_wgrad': f"conv_wgrad_{mi['kk']} (+ slab reduce)",
                     'conv_planes': 'conv_planes_mfma (DMA-fed forward AND stride-1 dgrad over pre-split operands; filter split, '
                                    'BN-stat fold kernels included)',
                     'wgrad_planes': 'wgrad_planes_mfma (DMA-fed wgrad over pre-split operands, transposed LDS reads; + slab reduce)'}[dom]
            # The roofline of the instruction stream that actually runs: a split mode spends `mfmas` dense 16-bit MFMAs per
            # fp32-grade product, so its ceiling in ALGORITHMIC flop/s is the 16-bit MFMA peak / mfmas; frac is then the
            # matrix-pipe utilisation at nominal clock (identical to executed flop/s over the 2.5 PFLOP/s dense peak).
            peak = mi['peak'] / mi['mfmas']
            tf = syms[dsym]['tflops']
            traffic, traffic_src = pmc_recorded('hbm_traffic', dsym)
            util, util_src = pmc_recorded('mfma_util', dsym)
            out['roofline'] = {'bound': 'mfma', 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak,
                               'traffic': traffic,
                               'kernel': dsym,
                               'measured': 'achieved / frac / avg_launch_ms / per_symbol: LIVE in this run (HIP events on the launch '
                                           'stream around ' + ('every conv launch' if all_events else 'every launch of the dominant kernel '
                                           '(--all-kernel-events: all conv launches)') + ' of the timed steps)',
                               'recorded': {'note': 'PMC counters cannot be read from inside the process: traffic and mfma_util_pmc are '
                                                    'RECORDED values of separate rocprofv3 --pmc passes of this same command, not '
                                                    'measurements of this run; null when no recording of this kernel symbol exists',
                                            'traffic_source': traffic_src, 'mfma_util_pmc': util, 'mfma_util_source': util_src},
                               'peak_note': mi['peak_note'],
                               'frac_of_dense_16bit_peak_algorithmic': tf / PEAK_BF16_MFMA_TFLOPS,
                               'vs_fp32_mfma_peak': tf / PEAK_F32_MFMA_TFLOPS,
                               'kernel_note': 'dominant kernel symbol by total time inside the timed steps; the HIP-event bracket '
                                              'around each launch also covers its filter split / slab-reduce / BN-stat fold '
                                              'helpers (a few % of the duration); family: ' + kname,
                               'mfma_pipe': {'instr': mi['instr'], 'executed_tflops': mi['mfmas'] * tf, 'peak': mi['peak'],
                                             'frac': mi['mfmas'] * tf / mi['peak']},
                               'avg_launch_ms': syms[dsym]['seconds'] / syms[dsym]['launches'] * 1e3,
                               'launches': syms[dsym]['launches'],
                               'all_conv_kernels': {k: {'tflops': round(v['tflops'], 2), 'ms_per_step': round(v['seconds'] / args.steps * 1e3, 2)}
                                                    for k, v in summ.items()},
                               'per_symbol': {k: {'tflops': round(v['tflops'], 2), 'avg_launch_ms': round(v['seconds'] / v['launches'] * 1e3, 4),
                                                  'launches': v['launches']}
                                              for k, v in sorted(syms.items(), key=lambda kv: -kv[1]['seconds'])}}
        if args.conv_table and timer.rec:
            with open(args.conv_table, 'w') as f:
                f.write(timer.table(args.steps, B) + '\n')
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(S, args.cpu_batch)
            # the HIP path on configs[0] beside it (same shape; parity at this shape: tests/test_gpu_round2.py config0 test)
            model.eval()
            x0 = recipe.randn((1, 3, 416, 416), 7).to(dev)
            with torch.no_grad():
                for _ in range(3):
                    model(x0)
                torch.cuda.synchronize(dev)
                t0 = time.time()
                for _ in range(20):
                    model(x0)
                torch.cuda.synchronize(dev)
            out['cpu_baseline']['config0_416_forward_images_per_sec_hip'] = 20.0 / (time.time() - t0)
            model.train()
        if world == 1 and not args.no_infer_leg and args.conv_mode == 'f16x2' and args.batch == 64:
            # BASELINE configs[1] on the same clock, AFTER the headline's timed region (it shares nothing with it: its own model
            # in eval mode, batch 32): eval forward + decode + postprocess for a few batches -> `inference`
            del loss
            ddp.zero_grad()
            torch.cuda.empty_cache()
            leg = infer_leg(args, steps=6, warmup=2)
            out['inference'] = {'metric': leg['metric'], 'value': leg['value'], 'unit': leg['unit'], 'ms_per_step': leg['ms_per_step'],
                                'steps': leg['steps'], 'warmup': leg['warmup'],
                                'forward_ms': leg['config']['forward_ms'], 'postprocess_ms': leg['config']['postprocess_ms'],
                                'survivors_per_img': leg['config']['survivors_per_img'], 'workload': leg['config']['workload'],
                                'roofline': leg['roofline']}
        quiet.emit(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
