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

Drop-in: the swap at main_amp.py:131 is one line,

    model = BucketedDDP(model)          # was: DDP(model, delay_allreduce=True)

and the reference's loop (yolo/engine/build.py:53-69: optimizer.zero_grad(); output = model(input);
loss.backward(); optimizer.step(); optimizer.zero_grad()) runs unchanged: `forward` re-arms the buckets and
heals the gradient slots (an `optimizer.zero_grad()` with torch's default set_to_none=True detaches
`p.grad` from the flat buffers), the first gradient hook of a backward queues an end-of-backward
callback that waits for the exchanges.  `zero_grad()`, `rearm()` and `finish_backward()` remain
callable (idempotent) for explicit control, e.g. gradient accumulation without exchange.
"""
import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ('flat', 'params', 'views', 'pending', 'work', 'launched', 'seen')

    def __init__(self, flat, params, views):
        self.flat, self.params, self.views = flat, params, views
        self.pending, self.work, self.launched, self.seen = 0, None, False, set()


def _align(n, a):
    return (n + a - 1) // a * a


class BucketedDDP(torch.nn.Module):

    def __init__(self, module, bucket_mb=25.0, process_group=None, broadcast=True, delay_allreduce=True, **_apex_kw):
        """delay_allreduce / other apex keyword arguments are accepted for call compatibility and ignored."""
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
        self._in_backward = False     # an end-of-backward callback is queued
        self._finished = True         # finish_backward() ran for the last backward
        self.stats = {'healed': 0, 'copied_in': 0}      # slots re-attached / stray gradients copied in (diagnostics)
        self._hooks = []
        for b in self.buckets:
            for p in b.params:
                hook = self._make_hook(b)
                self._hooks.append(p.register_post_accumulate_grad_hook(hook))
                # gradients produced outside autograd (yolov4_amd.ops: wgrad on the side stream) report here
                p._y4_grad_ready = (lambda h=hook, q=p: h(q))
                p._y4_ddp = self
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
    def _slot_view(flat, o, p):
        # the gradient of a KRSC (channels_last) filter keeps that memory layout inside the flat buffer
        g = flat[o:o + p.numel()]
        if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
            Co, Ci, kh, kw = p.shape
            return g.view(Co, kh, kw, Ci).permute(0, 3, 1, 2)
        return g.view(p.shape)

    @classmethod
    def _make_bucket(cls, params):
        dev, dt = params[0].device, params[0].dtype
        # every slot starts on a 16-byte boundary (the conv kernels write filter gradients with 16-B stores; the
        # three 255-element head biases would otherwise leave all later slots misaligned); the pad words stay zero
        offs, o = [], 0
        for p in params:
            offs.append(o)
            o = _align(o + p.numel(), 4)
        flat = torch.zeros(o, device=dev, dtype=dt)
        views = []
        for p, off in zip(params, offs):
            g = cls._slot_view(flat, off, p)
            p.grad = g
            views.append(g)
        return _Bucket(flat, list(params), views)

    def _make_hook(self, bucket):
        index = {id(p): i for i, p in enumerate(bucket.params)}

        def hook(param):
            # Once per parameter and backward.  A gradient written in place by a kernel is reported by hand
            # (param._y4_grad_ready); the Function then returns None for that input and torch (2.10) STILL runs the
            # post-accumulate hook of the leaf -- counted twice, a bucket would be exchanged before its last gradient
            # has been written.
            if id(param) in bucket.seen:
                return
            bucket.seen.add(id(param))
            if not self._in_backward:
                self._begin_backward()
            # a gradient that does not live in its slot (p.grad was None or replaced after forward): move it in
            g, v = param.grad, bucket.views[index[id(param)]]
            if g is not None and (g.data_ptr() != v.data_ptr() or g.stride() != v.stride()):
                with torch.no_grad():
                    v.copy_(g)
                param.grad = v
                self.stats['copied_in'] += 1
            bucket.pending -= 1
            if bucket.pending == 0 and not self.accumulating:
                self._launch(bucket)
        return hook

    # -- measurement aid: when each bucket's exchange is issued relative to the backward pass (bench.py)
    def record_timeline(self, on=True):
        self._timeline = {'begin': None, 'launch': [], 'end': None} if on else None

    def _mark(self, what, bucket=None):
        tl = getattr(self, '_timeline', None)
        if tl is None or not self.buckets or not self.buckets[0].flat.is_cuda:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if what == 'launch':
            tl['launch'].append((self.buckets.index(bucket), bucket.flat.numel() * bucket.flat.element_size(), ev))
        else:
            if what == 'begin':
                tl['launch'] = []
            tl[what] = ev

    def timeline(self):
        """[(bucket index, bytes, ms after the first gradient of the backward pass, ms before its end)] of the last
        backward pass; events are on the stream the collective is issued from (call after a device synchronize)."""
        tl = getattr(self, '_timeline', None)
        if not tl or tl['begin'] is None or tl['end'] is None:
            return []
        return [(i, nb, tl['begin'].elapsed_time(ev), ev.elapsed_time(tl['end'])) for i, nb, ev in tl['launch']]

    def _begin_backward(self):
        self._in_backward = True
        self._finished = False
        self._mark('begin')
        try:
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        except RuntimeError:
            # not inside a backward pass (gradient reported by hand): the caller runs finish_backward()
            self._in_backward = False

    def _end_of_backward(self):
        self._in_backward = False
        self.finish_backward()

    def _launch(self, b):
        timed = getattr(self, '_timeline', None) is not None
        if b.launched or not (self.use_dist or timed):
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
        self._mark('launch', b)               # completes when the bucket's last gradient has landed
        if self.use_dist:
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
    def _heal(self):
        """Re-attach every gradient slot that was detached from its flat buffer (p.grad set to None by
        optimizer.zero_grad(set_to_none=True), torch's default, or replaced by a foreign tensor).  A detached slot
        means "this gradient is zero": the slot is cleared and may then be written in place by the producing kernel."""
        for b in self.buckets:
            lost = [i for i, p in enumerate(b.params)
                    if p.grad is None or p.grad.data_ptr() != b.views[i].data_ptr() or p.grad.stride() != b.views[i].stride()]
            if not lost:
                continue
            with torch.no_grad():
                if len(lost) == len(b.params):
                    b.flat.zero_()
                else:
                    for i in lost:
                        b.views[i].zero_()
            for i in lost:
                p = b.params[i]
                p.grad = b.views[i]
                p._y4_grad_fresh = True
            self.stats['healed'] += len(lost)

    def zero_grad(self, set_to_none=False):
        """Clears the flat buckets (one memset each) and re-arms them; gradients stay views of the buckets."""
        for b in self.buckets:
            with torch.no_grad():
                b.flat.zero_()
            for i, p in enumerate(b.params):
                p.grad = b.views[i]
                # a zeroed slot may be written in place by the producing kernel (yolov4_amd.ops.ConvBNActFn)
                p._y4_grad_fresh = True
        self.rearm()

    def rearm(self):
        """Before every backward (forward() does it): gradients keep accumulating in the flat buckets across the
        micro-steps of an accumulation window (yolo/engine/build.py:56-69 steps every ACCUMULATION_STEPS)."""
        for b in self.buckets:
            b.pending, b.work, b.launched = len(b.params), None, False
            b.seen.clear()

    def forward(self, *a, **kw):
        if torch.is_grad_enabled() and self.module.training:
            self._heal()
            self.rearm()
        return self.module(*a, **kw)

    def finish_backward(self):
        """Waits for the bucket exchanges (stream-wise: the current stream waits for RCCL's) and applies the
        1/world average where the backend has no AVG.  Runs by itself at the end of every backward (autograd
        callback); calling it again afterwards is a no-op."""
        if self._finished:
            return
        self._finished = True
        if self.buckets and self.buckets[0].flat.is_cuda:
            from . import ops
            ops.join_side_stream()
        self._mark('end')
        if self.accumulating:                 # local accumulation only: exchange happens on the last micro-step
            return
        for b in self.buckets:
            if not b.launched:
                self._launch(b)               # parameters that received no gradient this step
            if b.work is not None:
                b.work.wait()
                if self.backend != 'nccl':
                    b.flat.div_(self.world)
                b.work = None
