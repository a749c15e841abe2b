"""Data-parallel gradient exchange: one flat fp32 bucket, one RCCL all-reduce per iteration.

The reference is single-GPU (MD2 README.md:149); this is new work.  One process
per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI; ``gloo`` on CPU for tests).  Every
trainable tensor's ``.grad`` is a view into ONE contiguous buffer (57.3 MB for ResNet-18 + decoder), so
the exchange is a single all-reduce; it is issued on a side stream right after backward and the
training loop only waits for it before ``optimizer.step()`` -- in the default (throughput) mode that wait
is placed AFTER the next iteration's attack has been enqueued, so the collective overlaps the attack.
xGMI is point-to-point (7 links x ~153 GB/s): a ring moves 2*(N-1)/N * 57.3 MB per GPU ~ 0.65 ms at
N=8 -- latency-, not bandwidth-bound, hence one bucket rather than many.
"""
import contextlib
import os

import torch
import torch.distributed as dist


def init_distributed(device_type="cuda"):
    """Initialise from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type == "cuda":
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if (world > 1 or os.environ.get("DMH_DIST_FORCE_INIT")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # nccl (= RCCL on ROCm) for GPUs; DMH_DIST_BACKEND=gloo lets several ranks share ONE GPU in tests
        backend = os.environ.get("DMH_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")
        kw = {"device_id": device} if backend == "nccl" else {}
        # generous: rank 0 may spend minutes compiling MIOpen kernels on a fresh box while the others wait in a barrier
        import datetime
        kw["timeout"] = datetime.timedelta(minutes=int(os.environ.get("DMH_DIST_TIMEOUT_MIN", "60")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def shorten_timeout(minutes=None):
    """After warm-up (rendezvous done, MIOpen's kernels built) a collective that does not finish within minutes is a hang,
    not a slow rank: bring the process group's timeout down from the generous start-up value so that a stuck rank fails the
    job in ``minutes`` (DMH_DIST_STEADY_TIMEOUT_MIN, default 10) instead of an hour.  No-op without a process group or
    where this torch build does not expose the setter."""
    if not dist.is_initialized():
        return False
    import datetime
    minutes = float(os.environ.get("DMH_DIST_STEADY_TIMEOUT_MIN", "10")) if minutes is None else minutes     # fractions: tests
    setter = getattr(torch.distributed.distributed_c10d, "_set_pg_timeout", None)
    why = "this torch build has no distributed_c10d._set_pg_timeout"
    if setter is not None:
        try:
            setter(datetime.timedelta(minutes=minutes), dist.group.WORLD)
            return True
        except Exception as e:      # a private API: say so instead of silently keeping the start-up timeout
            why = "%s: %s" % (type(e).__name__, e)
    if dist.get_rank() == 0:
        import sys
        print("[ddp] note: the process group keeps its start-up timeout (a stuck rank fails the job later than %g min): %s"
              % (minutes, why), file=sys.stderr, flush=True)
    return False


class GradBucket(object):
    """Flat gradient bucket over the parameters that can receive a gradient.

    ``params``: iterable of tensors (requires_grad).  Parameters that never get a gradient (the encoder's
    unused ``fc``, MD2/trainer.py:85) should be excluded by the caller.
    """

    def __init__(self, params, world_size=None, group=None, force_collective=False):
        """``force_collective``: issue the all-reduce even with one rank (a 1-rank RCCL communicator) -- lets a single-GPU
        box exercise the real nccl + side-stream + event path (tests/test_gpu_ddp.py)."""
        self.force = bool(force_collective)
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise RuntimeError("GradBucket: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dt)
        self._offsets = []
        off = 0
        for p in self.params:
            self._offsets.append(off)
            off += p.numel()
        self.attach()
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._work = None
        self._event = None
        # timing=True: HIP events around the collective on the side stream and around the consumer's wait for it, read by
        # timings() after a synchronize (bench.py, N > 1: overlap evidence without a profiler)
        self.timing = False
        self._timed = []

    def timings(self, clear=True):
        """Per all-reduce since the last call: ``all_reduce_ms`` = first to last instruction of the collective (+ the 1/N
        scaling) on the side stream -- stretched when its kernels wait for CUs beside the attack; ``wait_ms`` = how long
        the consuming stream (the optimiser's) stood at the event: ~0 when the collective had finished behind the attack.
        The caller synchronises first."""
        out = {"all_reduce_ms": [], "wait_ms": []}
        for e0, e1, w0, w1 in self._timed:
            if w1 is None:
                continue
            out["all_reduce_ms"].append(e0.elapsed_time(e1))
            out["wait_ms"].append(w0.elapsed_time(w1))
        if clear:
            self._timed = [t for t in self._timed if t[3] is None]
        return out

    def attach(self):
        """(Re-)point every parameter's ``.grad`` at its slice of the flat buffer."""
        for p, off in zip(self.params, self._offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)  # autograd accumulates in place into the view

    def check_attached(self):
        """Every ``.grad`` must still alias its slice: ``optimizer.zero_grad(set_to_none=True)``, ``module.to(memory_format=
        ...)`` or ``p.grad = None`` silently detach it, after which zero() and the all-reduce would act on a buffer
        autograd no longer writes to.  Pointer compare only -- no device work."""
        base, esz = self.flat.data_ptr(), self.flat.element_size()
        for p, off in zip(self.params, self._offsets):
            g = p.grad
            if g is None or g.data_ptr() != base + off * esz or not g.is_contiguous():
                raise RuntimeError("GradBucket: the .grad of a %s parameter no longer aliases the flat bucket (was "
                                   "zero_grad(set_to_none=True) / .to(memory_format=...) called after the bucket was "
                                   "built?); call bucket.attach() or build the bucket last" % (tuple(p.shape),))

    def zero(self):
        self.check_attached()
        self.flat.zero_()

    def release(self):
        """Before ``backward()``: detach every ``.grad`` so that autograd hands each parameter the gradient tensor its producer
        wrote (AccumulateGrad keeps it as it is) instead of adding it into the zeroed view -- one element-wise add kernel per
        parameter otherwise (93 launches, 1.0 ms of a config-2 step).  ``collect()`` puts the bucket back together."""
        for p in self.params:
            p.grad = None

    def collect(self):
        """After ``backward()`` that followed ``release()``: copy the gradients into their slices of the flat buffer (one
        multi-tensor copy; parameters that received none read as zero) and re-point every ``.grad`` at its slice."""
        views = [self.flat[off:off + p.numel()].view_as(p) for p, off in zip(self.params, self._offsets)]
        grads, dsts = [], []
        for p, v in zip(self.params, views):
            if p.grad is not None:
                grads.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
                dsts.append(v)
        if len(grads) < len(views):
            self.flat.zero_()
        if grads:
            torch._foreach_copy_(dsts, grads)
        for p, v in zip(self.params, views):
            p.grad = v

    @contextlib.contextmanager
    def released(self):
        """``with bucket.released(): loss.backward()`` -- release() before, collect() after; when backward raises the
        views are re-attached (attach(): pointer work only) so that every ``.grad`` aliases its slice again -- otherwise the
        next zero() / check_attached() fails with a message about zero_grad(set_to_none=True) that hides the real error --
        and the ORIGINAL exception propagates."""
        self.release()
        try:
            yield self
        except BaseException:
            # backward failed -- possibly with a device error, after which collect()'s own device work (zero_, the
            # multi-tensor copy) would raise too and REPLACE the error that matters.  Re-point the views only (no device
            # work): whatever gradients were produced are dropped, the bucket is structurally whole again.
            self.attach()
            raise
        else:
            self.collect()

    def start_all_reduce(self):
        """Enqueue sum-all-reduce + 1/N of the bucket; returns immediately."""
        if self.world <= 1 and not self.force:
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                e0 = None
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._work.wait()           # orders the side stream after the collective (no host block)
                self.flat.div_(self.world)
                self._event = torch.cuda.Event(enable_timing=self.timing)
                self._event.record(self.stream)
                if e0 is not None:
                    self._timed.append([e0, self._event, None, None])
        else:
            self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish_all_reduce(self):
        """Make the averaged gradients visible to the current stream (call before optimizer.step())."""
        if self.world <= 1 and not self.force:
            return
        if self.stream is not None:
            if self._event is not None:
                cur = torch.cuda.current_stream()
                rec = self._timed[-1] if (self._timed and self._timed[-1][1] is self._event) else None
                if rec is not None:
                    rec[2] = torch.cuda.Event(enable_timing=True)
                    rec[2].record(cur)
                cur.wait_event(self._event)
                if rec is not None:
                    rec[3] = torch.cuda.Event(enable_timing=True)
                    rec[3].record(cur)
                self._event = None
        elif self._work is not None:
            self._work.wait()
            self.flat.div_(self.world)
        self._work = None

    def all_reduce(self):
        self.start_all_reduce()
        self.finish_all_reduce()


def broadcast_parameters(modules, src=0, group=None):
    """Make every rank start from rank ``src``'s weights and buffers (one flat broadcast per module)."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return
    for m in modules:
        tensors = [p.data for p in m.parameters()] + [b.data for b in m.buffers() if b.dtype.is_floating_point]
        if not tensors:
            continue
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def average_buffers(modules, group=None):
    """Average the floating-point buffers (BatchNorm running statistics) over the ranks: BatchNorm stays per rank
    during training (no SyncBN), so without this the rank-0 checkpoint and the eval-mode attack / val() would carry one
    shard's statistics only.  Collective: every rank must call it (end of each epoch)."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return
    bufs = [b for m in modules for b in m.buffers() if b.dtype.is_floating_point]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    off = 0
    for b in bufs:
        n = b.numel()
        b.copy_(flat[off:off + n].view_as(b))
        off += n
