"""Data-parallel gradient exchange over RCCL/xGMI -- interface of
src/training/train_distributed.py:44-60, 97-149 (init_distributed, reduce_tensor,
apply_gradient_allreduce).

The reference flattens all 41.4 M gradients into one buffer AFTER backward, runs a
single blocking all-reduce, divides, and copies everything back (two extra
full-gradient passes over HBM, no overlap).  Here, one process per GPU:
  * parameters and gradients live permanently in flat fp32 buffers (``p.data`` / ``p.grad``
    are views; a bucket is a contiguous slice), so there is no flatten and no copy-back,
    and the optimizer section runs on the same buffers (training/flat_optim.py);
  * a bucket's all-reduce is issued from a post-accumulate hook as soon as its last
    gradient has been produced, on the communicator's own stream, overlapping the
    rest of backward; buckets are filled in reverse registration order, which is
    the order backward produces gradients (decoder first);
  * averaging uses ReduceOp.AVG on RCCL (SUM + scale on gloo); backward's end
    callback only waits for the outstanding handles.
xGMI is point-to-point (7 links per GPU), so the default 32 MiB bucket keeps each
collective large enough to be link-bandwidth bound rather than latency bound while
still giving ~5 overlappable pieces for the 165 MB of E8 gradients.
"""
import os

import torch
import torch.distributed as dist
from torch.autograd import Variable


def reduce_tensor(tensor, num_gpus):
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= num_gpus
    return rt


def init_distributed(rank, num_gpus, group_name=None, dist_backend="nccl", dist_url="tcp://127.0.0.1:54321"):
    """One process per GPU.  ``dist_backend="nccl"`` is RCCL on ROCm; "gloo" runs on CPU (tests)."""
    if dist_backend == "nccl":
        assert torch.cuda.is_available(), "Distributed mode on the nccl/RCCL backend requires a GPU."
        torch.cuda.set_device(rank % torch.cuda.device_count())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group(dist_backend, init_method=dist_url, world_size=num_gpus, rank=rank)


class GradBuckets:
    """Flat gradient storage + overlapped all-reduce for one module.

    Parameters and gradients are views of two flat fp32 buffers (training/flat_optim.py::FlatParams, laid out in
    gradient-ready order); a bucket is a contiguous slice of the gradient buffer.  ``require_sync = False`` skips the
    exchange for one backward (gradient accumulation: the reference all-reduces on every micro-step,
    src/training/train_distributed.py:145-148 -- correct but wasteful; the sum of the micro-step gradients is
    exchanged once at the accumulation boundary here).  Without an initialised process group the class is only the
    flat storage (world size 1)."""

    def __init__(self, module, bucket_bytes=32 << 20, process_group=None, flat=None):
        from .flat_optim import FlatParams
        self.group = process_group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self.use_avg = self.distributed and dist.get_backend(process_group) == "nccl"
        # a gradient exchange takes place: several ranks -- or ONE rank of an initialised group with
        # CUM_EXCHANGE_ALONE=1, which runs every collective of the data-parallel step (bucketed all-reduce in the eager
        # backward, whole-buffer all-reduce between the two captured graphs) against the real backend on a one-GPU box
        self.exchanging = self.world > 1 or (self.distributed and os.environ.get("CUM_EXCHANGE_ALONE") == "1")
        self.flat = flat if flat is not None else FlatParams(module)
        self.params = self.flat.params
        self.require_sync = True
        self.buckets = []          # [flat gradient slice, [params], pending count]
        self.where = {}            # id(param) -> (bucket index, gradient view)
        self.handles = []
        self._armed = False
        for start, end, members in self.flat.slices(bucket_bytes):
            plist = [self.params[i] for i in members]
            for i in members:
                self.where[id(self.params[i])] = (len(self.buckets), self.flat.grad_views[i])
            self.buckets.append([self.flat.grad[start:end], plist, len(plist)])
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)
        self.flat.on_write = self._written
        # producers that can hand over part of their gradients before they are done with all of them do so when there is
        # an exchange to overlap with (network/convstack.py EncoderStack.backward)
        self.flat.early_announce = self.exchanging

    def _written(self, params):
        """Gradients a kernel wrote straight into the flat buffer (FlatParams.wrote): same bookkeeping as the hook."""
        for p in params:
            self._ready(p)

    def zero_grad(self):
        self.flat.zero_grad()

    def _hook(self, p):
        idx, view = self.where[id(p)]
        if p.grad is not view:
            # someone replaced .grad (e.g. zero_grad(set_to_none=True)): fold it back into the bucket
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            p.grad = view
        self.flat.fresh[self.flat.index[id(p)]] = False
        self._ready(p)

    def _ready(self, p):
        idx = self.where[id(p)][0]
        if not self.exchanging or not self.require_sync:
            return
        if not self._armed:
            self._armed = True
            self._seen = set()
            for b in self.buckets:
                b[2] = len(b[1])
            Variable._execution_engine.queue_callback(self._finish)
        # a parameter counts once per backward: autograd runs the post-accumulate hook even for the None gradient a
        # Function returns after it has written the real one into the flat buffer itself (gradient sink)
        if id(p) in self._seen:
            return
        self._seen.add(id(p))
        b = self.buckets[idx]
        b[2] -= 1
        if b[2] == 0:
            self._launch(b[0])

    def exchange_all(self):
        """Average the WHOLE flat gradient buffer over the ranks with one collective on the current stream (the form the
        captured train step uses between its two graphs: 165.5 MB at E8 in one call -- xGMI is point-to-point, a ring
        is per-link bound, so one large collective beats five 32 MiB ones when nothing overlaps them anyway)."""
        if not self.exchanging:
            return
        g = self.flat.grad
        if self.use_avg:
            dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(g, group=self.group)
            g /= self.world

    def exchange_range(self, start, end):
        """Average flat.grad[start:end] over the ranks, asynchronously: returns a wait() callable.  The collective is
        ordered behind the work already enqueued on the current stream and runs beside what is enqueued after this
        call (the process group's own stream on RCCL)."""
        if not self.exchanging or end <= start:
            return lambda: None
        g = self.flat.grad[start:end]
        if self.use_avg:
            h = dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            return h.wait

        h = dist.all_reduce(g, group=self.group, async_op=True)

        def wait():
            h.wait()
            g.div_(self.world)
        return wait

    def all_ranks_agree(self, ok):
        """True iff ``ok`` holds on every rank (an eager MIN all-reduce of one flag; call it outside any capture, at
        a point every rank reaches at the same step)."""
        if not self.distributed:
            return bool(ok)
        dev = self.flat.grad.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def _launch(self, flat):
        if self.use_avg:
            self.handles.append((dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:
            self.handles.append((dist.all_reduce(flat, group=self.group, async_op=True), flat))

    def _finish(self):
        # buckets whose parameters did not all receive a gradient this pass are still reduced,
        # so that every rank issues the same sequence of collectives
        for b in self.buckets:
            if b[2] != 0:
                self._launch(b[0])
                b[2] = 0
        for h, flat in self.handles:
            h.wait()
            if flat is not None:
                flat /= self.world
        self.handles = []
        self._armed = False


def apply_gradient_allreduce(module, bucket_bytes=32 << 20):
    """Broadcast rank 0's parameters/buffers, then all-reduce (average) gradients during every
    backward.  Does not change the module's class; returns the module (reference contract).
    The bucket manager is exposed as ``module.grad_buckets`` (use its zero_grad())."""
    with torch.no_grad():
        for t in module.state_dict().values():
            if torch.is_tensor(t):
                dist.broadcast(t, 0)
    module.grad_buckets = GradBuckets(module, bucket_bytes)
    return module
