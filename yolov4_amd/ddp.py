# -*- coding: utf-8 -*-
"""Data-parallel gradient exchange for one node of MI355X GPUs (RCCL over xGMI).

Replaces apex.parallel.DistributedDataParallel(model, delay_allreduce=True) at
main_amp.py:126-131 of the reference: same result (parameters broadcast from rank 0
at wrap time; after every backward each gradient is the SUM over ranks divided by
world_size), different schedule.  apex flattens all 259.5 MB of gradients and issues
one all-reduce after backward has finished; here gradients live in ~25 MB flat buckets
filled in reverse execution order (head -> neck -> stage5 ... stem), and a bucket's
all-reduce is issued on RCCL's own stream as soon as its last wgrad has been
enqueued, so the exchange runs under the remaining dgrad/wgrad kernels.  One process
per GPU, `torch.distributed` backend "nccl" (= RCCL on ROCm); "gloo" is supported for
the CPU tests of the bucket logic.
"""
import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ('flat', 'params', 'pending', 'work', 'launched')

    def __init__(self, flat, params):
        self.flat, self.params = flat, params
        self.pending, self.work, self.launched = 0, None, False


class BucketedDDP(torch.nn.Module):

    def __init__(self, module, bucket_mb=25.0, process_group=None, broadcast=True):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        self.backend = dist.get_backend(self.pg) if dist.is_initialized() else None
        # collectives run whenever a process group exists (also at world_size 1: exercises the RCCL path)
        self.use_dist = dist.is_initialized()
        if broadcast and self.use_dist:
            self._broadcast_state()
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()                       # backward produces gradients roughly in this order
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets, cur, n = [], [], 0
        for p in params:
            cur.append(p)
            n += p.numel()
            if n >= cap:
                self.buckets.append(self._make_bucket(cur))
                cur, n = [], 0
        if cur:
            self.buckets.append(self._make_bucket(cur))
        self.accumulating = False     # set True for all but the last micro-step of an accumulation window
        self._hooks = []
        for b in self.buckets:
            for p in b.params:
                hook = self._make_hook(b)
                self._hooks.append(p.register_post_accumulate_grad_hook(hook))
                # gradients produced outside autograd (yolov4_amd.ops: wgrad on the side stream) report here
                p._y4_grad_ready = (lambda h=hook, q=p: h(q))
        self.zero_grad()

    # -- setup
    def _broadcast_state(self):
        with torch.no_grad():
            tensors = [t for t in list(self.module.parameters()) + list(self.module.buffers())]
            for dtype in sorted({t.dtype for t in tensors}, key=str):
                group = [t for t in tensors if t.dtype == dtype]
                flat = torch.cat([t.reshape(-1) for t in group])
                dist.broadcast(flat, 0, group=self.pg)
                o = 0
                for t in group:
                    t.copy_(flat[o:o + t.numel()].view(t.shape))
                    o += t.numel()

    @staticmethod
    def _make_bucket(params):
        dev, dt = params[0].device, params[0].dtype
        flat = torch.zeros(sum(p.numel() for p in params), device=dev, dtype=dt)
        o = 0
        for p in params:
            # the gradient of a KRSC (channels_last) filter keeps that memory layout inside the flat buffer
            g = flat[o:o + p.numel()]
            if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
                Co, Ci, kh, kw = p.shape
                g = g.view(Co, kh, kw, Ci).permute(0, 3, 1, 2)
            else:
                g = g.view(p.shape)
            p.grad = g
            o += p.numel()
        return _Bucket(flat, list(params))

    def _make_hook(self, bucket):
        def hook(_param):
            bucket.pending -= 1
            if bucket.pending == 0 and not self.accumulating:
                self._launch(bucket)
        return hook

    def _launch(self, b):
        if b.launched or not self.use_dist:
            b.launched = True
            return
        op = dist.ReduceOp.AVG if self.backend == 'nccl' else dist.ReduceOp.SUM
        if b.flat.is_cuda:
            # gradients of one bucket are written from two streams (main: BN/bias grads, side: filter grads):
            # the collective is ordered after both
            cur = torch.cuda.current_stream(b.flat.device)
            for st in self._producer_streams(b.flat.device):
                if st != cur:
                    cur.wait_stream(st)
        b.work = dist.all_reduce(b.flat, op=op, group=self.pg, async_op=True)
        b.launched = True

    @staticmethod
    def _producer_streams(device):
        sts = [torch.cuda.default_stream(device)]
        try:
            from . import ops
            sts.extend(st for st in ops._ASYNC['streams'].values() if st.device == device)
        except Exception:
            pass
        return sts

    # -- per step
    def zero_grad(self, set_to_none=False):
        for b in self.buckets:
            b.flat.zero_()
            for p in b.params:
                # a zeroed slot may be written in place by the producing kernel (yolov4_amd.ops.ConvBNActFn)
                p._y4_grad_fresh = p.grad is not None
        self.rearm()

    def rearm(self):
        """Before every backward of an accumulation window after the first (gradients keep accumulating
        in the flat buckets; yolo/engine/build.py:56-69 steps every ACCUMULATION_STEPS micro-batches)."""
        for b in self.buckets:
            b.pending, b.work, b.launched = len(b.params), None, False

    def forward(self, *a, **kw):
        return self.module(*a, **kw)

    def finish_backward(self):
        """Call after loss.backward(): waits for the bucket exchanges (stream-wise: the current stream
        waits for RCCL's) and applies the 1/world average where the backend has no AVG."""
        if self.buckets and self.buckets[0].flat.is_cuda:
            from . import ops
            ops.join_side_stream()
        if self.accumulating:                 # local accumulation only: exchange happens on the last micro-step
            return
        for b in self.buckets:
            if not b.launched:
                self._launch(b)               # parameters that received no gradient this step
            if b.work is not None:
                b.work.wait()
                if self.backend != 'nccl':
                    b.flat.div_(self.world)
